"""Loader / builder for libremfx_hip.so (the C-ABI boundary, include/remfx_hip.h).

The library is built IN-TREE (remfx_amd/_C/libremfx_hip.so) with
``hipcc --offload-arch=gfx950`` so it travels to the GPU box with the repo
snapshot.  There is deliberately NO fallback: if the library is missing or a
symbol declared in the header is absent, importing the op layer raises.
"""
import ctypes as C
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIBDIR = os.path.join(_HERE, "_C")
LIBPATH = os.environ.get("RFX_LIBPATH_DEV") or os.path.join(LIBDIR, "libremfx_hip.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
               "-I" + INCLUDE, "-I" + CSRC]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared library (object per
    file so unchanged files are not recompiled)."""
    if not force and not _stale():
        return LIBPATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    hdr_t = max([os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "*.h")) +
                 glob.glob(os.path.join(INCLUDE, "*.h"))] or [0])
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_t):
            continue
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBPATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return LIBPATH


# ---- ctypes mirrors of the header structs ----------------------------------------
class KtabEntry(C.Structure):
    _fields_ = [("off", C.c_int32), ("da", C.c_int32), ("db", C.c_int32), ("flags", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "M", "K", "OA", "OB", "IA", "IB", "SA", "SB", "Mpad",
                                         "Kpad", "out_a0", "out_b0", "out_sa", "out_sb", "R", "mg_log", "mg_axis", "mg_len", "mg_off",
                                         "Kpad_t", "gpt", "ntaps", "gpt2")] + \
               [(n, C.c_int64) for n in ("in_ns", "in_as", "in_bs", "out_ns", "out_cs", "out_as", "out_bs", "in_cs", "in_extent")] + \
               [("in_bf16", C.c_int32), ("out_bf16", C.c_int32)] + \
               [(n, C.c_int32) for n in ("halo_nt", "halo_rows", "halo_w", "halo_da0", "halo_db0", "halo_pad")]


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("act", C.c_int32), ("act_param", C.c_void_p),
                ("res", C.c_void_p), ("res_ns", C.c_int64), ("res_cs", C.c_int64),
                ("res_as", C.c_int64), ("res_bs", C.c_int64), ("act2", C.c_int32), ("bwd", C.c_int32),
                ("gparam", C.c_void_p), ("stat_sums", C.c_void_p), ("stat_slots", C.c_int32),
                ("glu_out", C.c_void_p), ("glu_ns", C.c_int64)]


class StftDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("R", "T", "n_fft", "hop", "win", "bins", "frame0", "frames_out",
                                         "mode", "extra_pad_l", "extra_pad_r", "in_mode", "in_offset",
                                         "herm")] + \
               [(n, C.c_float) for n in ("scale", "eps", "alpha")] + [("accum", C.c_int32)]


class ClTensor(C.Structure):
    """rfx_cl_tensor: a channels-last bf16 operand [N][A][B][bs] (include/remfx_hip.h)."""
    _fields_ = [("p", C.c_void_p), ("ns", C.c_int64), ("as_", C.c_int64), ("bs", C.c_int32), ("c0", C.c_int32)]


class ClConvDesc(C.Structure):
    _fields_ = [("inp", ClTensor)] + \
               [(n, C.c_int32) for n in ("N", "IA", "IB", "OA", "OB", "SA", "NTR", "NCH", "NTC", "KS", "da0", "da_step", "db0",
                                         "db_step", "wrapb")] + \
               [("apack", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("M", "BM", "mode", "G", "g_off", "OAo", "Co")] + \
               [("bias", C.c_void_p), ("rowadd", C.c_void_p), ("out0", ClTensor), ("out1", ClTensor), ("aux0", ClTensor), ("res", ClTensor),
                ("cm_out", C.c_void_p), ("cm_ns", C.c_int64), ("cm_cs", C.c_int64), ("cm_as", C.c_int64), ("cm_fold", C.c_int32)]


class ClDconvDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "gy", "y", "a", "hpre", "stats", "dz", "dh", "partial", "w1p", "w2p", "w2dp", "w1dp",
                                          "b1", "g1w", "g1b", "b2", "g2w", "g2b", "scale")] + \
               [(n, C.c_int32) for n in ("S", "C", "H", "dil", "grid", "x_or_gy_ok")] + [("eps", C.c_float), ("TPS", C.c_int32), ("tsum", C.c_void_p), ("sums", C.c_void_p), ("pg_dst", C.c_void_p * 5)]


class ClWgradDesc(C.Structure):
    _fields_ = [("p", ClTensor), ("q", ClTensor)] + \
               [(n, C.c_int32) for n in ("N", "OA", "IA", "B", "SA", "da0", "NTR", "NTC", "db0", "db_step", "M", "Cq", "CW", "RW", "WK",
                                         "S", "ahead", "bias", "PW")] + \
               [("ws", C.c_void_p)]


_P, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
# name -> argtypes; every symbol include/remfx_hip.h declares must be listed here
SIGNATURES = {
    "rfx_abi_version": [],
    "rfx_gemm_pick_r": [_I32, _I32],
    "rfx_pack_a": [_P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "rfx_unpack_add": [_P, _P, _I64, _I32, _I32, _I32, _P, _I32, _P],
    "rfx_unpack_add_bias": [_P, _P, _I64, _I32, _I32, _I32, _P, _I32, _P, _I32, _P],
    "rfx_unpack_set": [_P, _P, _I64, _I32, _I32, _I32, _P, _I32, _P],
    "rfx_unpack_col": [_P, _I32, _I32, _I32, _I32, _P, _P],
    "rfx_gemm_fwd": [C.POINTER(GemmDesc), _P, _P, _P, _P, C.POINTER(Epilogue), _P, _P, _I32, _I32, _P, _I32, _P],
    "rfx_gemm_fwd_variant": [C.POINTER(GemmDesc), C.POINTER(Epilogue), _I32, _I32],
    "rfx_gemm_wgrad": [C.POINTER(GemmDesc), _P, _P, _P, _P, _I64, C.POINTER(C.c_int32), _I32, _P],
    "rfx_fft_analysis": [C.POINTER(StftDesc), _P, _P, _P, _P, _P],
    "rfx_fft_synthesis_ws": [C.POINTER(StftDesc)],
    "rfx_fft_synthesis": [C.POINTER(StftDesc), _P, _P, _P, _P, _P, _P],
    "rfx_stft_loss_reduce": [_P, _P, _I32, _I64, C.c_float, _P, _P, _P],
    "rfx_stft_loss_grad": [_P, _P, _I32, _I64, C.c_float, _P, C.c_float, C.c_float, _P, _P, _P],
    "rfx_stft_loss_grad_m": [_P, _P, _I32, _I64, C.c_float, _P, C.c_float, C.c_float, _P, _P, _P],
    "rfx_stft_pair_loss_ws": [C.POINTER(StftDesc)],
    "rfx_stft_pair_loss": [C.POINTER(StftDesc), _P, _P, _P, C.c_float, _P, _P, _P, _P, _P],
    "rfx_fft_synthesis_lossgrad": [C.POINTER(StftDesc), _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P, _P],
    "rfx_l1_grad": [_P, _P, _I64, C.c_float, _P, _P, _P],
    "rfx_sisdr_sums": [_P, _P, _I32, _I64, _I64, _I64, _P, _P, _P],
    "rfx_sisdr_finish": [_P, _I32, _I64, _I32, C.c_double, _P, _P],
    "rfx_mrstft_combine": [_P, _P, _I32, _I32, _I32, _P, _P],
    "rfx_zero": [_P, _I64, _P],
    "rfx_sumsq": [_P, _I64, _P, _P, _P],
    "rfx_clip_coef": [_P, C.c_float, C.c_float, _P, _P, _P],
    "rfx_adamw_step": [_P, _P, _P, _P, _I64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I32, _P, _P],
    "rfx_groupnorm_stat_chunks": [_I32, _I32, _I32],
    "rfx_norm_bwd_work_floats": [_I32, _I32, _I32, _I32],
    "rfx_batchnorm_stat_slots": [_I32, _I32],
    "rfx_groupnorm_fwd": [_P, _P, _P, _I32, _I32, _I32, _I32, C.c_float, _I32, _P, _P, _P, _I32, _P, _P, _P, _P],
    "rfx_groupnorm_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P],
    "rfx_groupnorm_fwd_x16": [_P, _P, _P, _I32, _I32, _I32, _I32, C.c_float, _I32, _P, _P, _P, _I32, _P, _P, _P, _P],
    "rfx_groupnorm_bwd_x16": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P],
    "rfx_batchnorm_fwd": [_P, _P, _P, _I32, _I32, _I32, C.c_float, _I32, _I32, _P, _P, _P, _P, _P],
    "rfx_batchnorm_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "rfx_avgpool2d_fwd": [_P, _P, _I64, _I32, _I32, _I32, _I32, _P],
    "rfx_avgpool2d_bwd": [_P, _P, _I64, _I32, _I32, _I32, _I32, _P],
    "rfx_cplx_slots": [_I32, _I64],
    "rfx_cplx_moments": [_P, _I32, _I32, _I64, _P, _P, _P],
    "rfx_cplx_moments_bwd": [_P, _P, _I32, _I32, _I64, _P, _P],
    "rfx_cplx_coef_fwd": [_P, C.c_double, _P, _P, _P, _P, _P, _P, C.c_float, _I32, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P],
    "rfx_cplx_coef_bwd": [_P, C.c_double, _P, _P, _P, _P, _P, _P, C.c_float, _I32, _P, _P, _P, _P],
    "rfx_cplx_affine_act_fwd": [_P, _P, _I32, _I32, _I64, C.c_float, _P, _I64, _I64, _P],
    "rfx_cplx_affine_act_bwd": [_P, _P, _P, _I64, _I64, _I32, _I32, _I64, C.c_float, _P, _P, _P, _P],
    "rfx_bound_mask_fwd": [_P, _P, _P, _I32, _I64, _I64, _I64, _I64, _P],
    "rfx_bound_mask_bwd": [_P, _P, _P, _P, _I32, _I64, _I64, _I64, _I64, _I64, _P],
    "rfx_phase_mask_fwd": [_P, _P, _P, _I64, _P],
    "rfx_phase_mask_bwd": [_P, _P, _P, _I64, _P],
    "rfx_glu_fwd": [_P, _P, _I64, _I64, _I64, _P],
    "rfx_glu_bwd": [_P, _P, _P, _I64, _I64, _I64, _P],
    "rfx_glu_bwd_bf16": [_P, _P, _P, _I64, _I64, _I64, _P],
    "rfx_act_fwd": [_P, _P, _I64, _I32, _P],
    "rfx_act_bwd": [_P, _P, _P, _I64, _I32, _P],
    "rfx_act_add_fwd": [_P, _P, _P, _I64, _I32, _P],
    "rfx_act_rows": [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _P],
    "rfx_act_rows16": [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _P],
    "rfx_mul": [_P, _P, _P, _I64, _P],
    "rfx_prelu_fwd": [_P, _P, _P, _I64, _I64, _I64, _P],
    "rfx_prelu_bwd": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "rfx_l1_sum": [_P, _P, _I64, _P, C.c_float, _P, _P],
    "rfx_add_bcast": [_P, _P, _P, _I64, _I32, _I32, _I32, _I64, _I64, _I64, _I64, C.c_float, _P],
    "rfx_localstate_fwd": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P],
    "rfx_localstate_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "rfx_dconv_layer_ok": [_I32, _I32, _I32],
    "rfx_dconv_layer_bwd_rows": [_I32],
    "rfx_dconv_layer_bwd": [_P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, _P],
    "rfx_dconv_layer_fwd": [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, _P],
    "rfx_fx_distortion": [_P, _P, _I32, _I64, _P, _P],
    "rfx_fx_delay": [_P, _P, _I32, _I64, _P, _P, _P, _P],
    "rfx_fx_chorus": [_P, _P, _I32, _I64, C.c_float, _P, _P, _P, _P, _P, _P],
    "rfx_fx_compressor": [_P, _P, _P, _I32, _I64, _P, _P, _P, _P, _P],
    "rfx_fx_reverb": [_P, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P],
    "rfx_fx_loudness": [_P, _I32, _I64, _I32, _I32, _I32, _I32, C.c_double, _P, C.c_float, _P, _P, _P, _P],
    "rfx_fx_scale": [_P, _P, _I32, _I64, _P, _P],
    "rfx_localstate_gen_fwd": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P],
    "rfx_localstate_gen_bwd": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "rfx_mha_fwd": [_P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P],
    "rfx_mha_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P],
    "rfx_localstate_mfma_ok": [_I32, _I32, _I32, _I32, _I32],
    "rfx_localstate_mfma_fwd": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "rfx_localstate_mfma_bwd": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P],
    "rfx_blstm_frames": [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "rfx_span_mask": [_P, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "rfx_dropout": [_P, _P, _I64, C.c_float, C.c_uint64, _P],
    "rfx_row_moments_slots": [_I64],
    "rfx_row_moments": [_P, _I32, _I64, _P, _P, _P, C.c_float, _P, _P, _P],
    "rfx_row_affine": [_P, _P, _P, _P, _I32, _I64, _P],
    "rfx_lstm_pack_bytes": [_I32],
    "rfx_lstm_pack": [_P, _I32, _P, _P],
    "rfx_lstm_cat_params": [_P, _P, _P, _P, _P, _P, _I32, _I32, _P, _P, _P],
    "rfx_lstm_grad_scatter": [_P, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P],
    "rfx_lstm_local": [_I32, _I32, _I32, _I32],
    "rfx_lstm_pack_local": [_P, _I32, _P, _P],
    "rfx_lstm_ws_bytes": [_I32],
    "rfx_lstm_fwd": [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _I32, _P],
    "rfx_lstm_bwd": [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _I32, _P],
    "rfx_lstm_set_local": [_I32, _I32],
    "rfx_channel_sum_ws": [_P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64],
    "rfx_channel_sum": [_P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _P, _P, _P],
    "rfx_cl_conv": [C.POINTER(ClConvDesc), _P],
    "rfx_cl_pack": [_P, _P, _I64, _P, _P],
    "rfx_cl_wgrad_ws_floats": [_P],
    "rfx_cl_wgrad": [_P, _P],
    "rfx_cl_wgrad_reduce": [_P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I64, _P, _I32, _P],
    "rfx_cl_from_cm": [_P, _I32, _I64, _I64, _I64, _I32, _I32, _I32, _I32, C.POINTER(ClTensor), C.POINTER(ClTensor), C.POINTER(ClTensor),
                       _I32, _P],
    "rfx_cl_rowsum": [C.POINTER(ClTensor), _I32, _I32, _I32, _I32, _I32, C.c_float, _P, _P, _I32, _P],
    "rfx_cl_to_cm": [C.POINTER(ClTensor), _I32, _I32, _I32, _I32, _P, _I32, _I64, _I64, _I64, _P, _P],
    "rfx_cl_dgelu": [_P, _P, _P, _I64, _P],
    "rfx_cl_dglu": [_P, _P, _P, _I64, _I32, _P],
    "rfx_cl_im2col_s4": [_P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "rfx_cl_im2col_fm": [_P, _P, _P, _I32, _I32, _I32, _P, _P],
    "rfx_fm_cm_affine": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "rfx_cl_dconv_ok": [_I32, _I32, _I32, _I32],
    "rfx_cl_dconv_fwd": [_P, _P],
    "rfx_cl_dconv_bwd": [_P, _P, _P],
}

_RET64 = {"rfx_cl_wgrad_ws_floats", "rfx_cplx_slots", "rfx_norm_bwd_work_floats", "rfx_stft_pair_loss_ws", "rfx_channel_sum_ws", "rfx_fft_synthesis_ws"}
_lib = None


def lib():
    """Load the shared library (never builds implicitly on a box without hipcc sources
    newer than the .so).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime instance that is
    # resident before our library is dlopen()ed, otherwise two runtimes coexist and
    # launches on torch's streams fail with hipErrorNoDevice.
    import torch  # noqa: F401
    if not os.path.exists(LIBPATH):
        raise RuntimeError(
            f"{LIBPATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  remfx_amd has no CPU fallback.")
    L = C.CDLL(LIBPATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the export is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name in _RET64 else C.c_int
    if L.rfx_abi_version() != 1:
        raise RuntimeError("libremfx_hip ABI version mismatch")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")
