/*
 * remfx_hip.h -- C ABI of libremfx_hip.so, the MI355X (gfx950) kernels behind the
 * RemFX effect-removal hot path.
 *
 * The reference (mhrice/RemFx) has no FFI of its own: its hot path is a sequence
 * of ATen op calls issued from Python.  Each entry point below
 * therefore cites the reference call site(s) whose ATen ops it replaces.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes only; no torch types.
 *   - the CALLER owns every buffer (inputs, outputs, workspaces); the library
 *     never allocates, frees, synchronises or changes the current device.
 *   - all work is enqueued on the hipStream_t passed as `stream` (void*).
 *   - return value: 0 on success, negative on bad arguments (-1) or a failed
 *     launch (-2 - hipError).  No exceptions cross the ABI.
 *   - tensors are fp32 unless stated; strides are in ELEMENTS.
 */
#ifndef REMFX_HIP_H
#define REMFX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFX_ABI_VERSION 1

/* ---- activations usable in fused epilogues ---------------------------------- */
enum rfx_act {
  RFX_ACT_NONE = 0,
  RFX_ACT_RELU = 1,   /* classifier.py:271-272 */
  RFX_ACT_GELU = 2,   /* exact erf GELU: HDemucs enc/dec, DConv */
  RFX_ACT_TANH = 3,   /* tcn.py:129 */
  RFX_ACT_PRELU = 4,  /* per-output-channel slope: tcn.py:47,51 */
  RFX_ACT_LEAKY = 5,  /* slope 0.01: DCUNet */
  RFX_ACT_SIGMOID = 6 /* classifier.py:231 */
};

/* ---- gather-GEMM descriptor -------------------------------------------------
 * One descriptor describes  Out[n, m, a', b'] = epi( sum_k A[k][m] * In(n, k, a, b) )
 * for a in [0,OA), b in [0,OB):
 *   In(n,k,a,b) = in[n*in_ns + ktab[k].off + (a*SA)*in_as + (b*SB)*in_bs]
 *                 if 0 <= a*SA + ktab[k].da < IA and 0 <= b*SB + ktab[k].db < IB, else 0
 *                 (ktab[k].off already contains channel*in_cs + da*in_as + db*in_bs)
 *   a' = a*out_sa + out_a0,  b' = b*out_sb + out_b0
 * Every convolution of the hot path (Conv1d/2d forward, their input gradients,
 * ConvTranspose1d/2d forward and input gradients, one descriptor per stride
 * phase) is expressed this way by the host-side planner (remfx_amd/convplan.py).
 * Replaces: F.conv1d tcn.py:50,54,129; HDemucs / DCUNet / Cnn14 conv stacks
 * (models.py:319,358; classifier.py:271-272).
 */
typedef struct rfx_ktab_entry {
  int32_t off;   /* element offset relative to the sample base */
  int32_t da;    /* tap displacement along a (for the bounds test) */
  int32_t db;    /* tap displacement along b */
  int32_t flags; /* bit0: constant-one column (bias-gradient row in wgrad) */
} rfx_ktab_entry;

typedef struct rfx_gemm_desc {
  int32_t N, M, K;       /* batch, output rows (channels), reduction length */
  int32_t OA, OB;        /* output position grid */
  int32_t IA, IB;        /* input extent for the bounds test */
  int32_t SA, SB;        /* input position stride */
  int32_t Mpad, Kpad;    /* packed-A geometry: A is [Kpad][Mpad], Mpad%4==0, Kpad%16==0 */
  int32_t out_a0, out_b0, out_sa, out_sb;
  int32_t R;             /* channel tiles (32 rows) per wave chosen by the planner: rfx_gemm_pick_r(M, K); 0 = thin path
                          * (M <= 8 only).  A phase-merged plan with M <= 8 sets R = 1, Mpad = 32: the thin kernels have no
                          * merged store, and one MFMA launch reads the operand once instead of once per phase. */
  /* Phase-merged output (mg_log > 0): the G = 2^mg_log stride phases of a transposed convolution (or of the input
   * gradient of a strided convolution) run as ONE GEMM with rows m = channel*G + phase, so the gathers are shared by G
   * times more MFMA work.  Row m, position index i on axis mg_axis (0 = A, 1 = B) is stored at channel m >> mg_log,
   * axis index i*G + (m & (G-1)) + mg_off if that lies in [0, mg_len); bias is indexed by the channel.  Epilogue
   * options other than bias / act / res are not available in this mode, nor is the thin (M <= 8) path. */
  int32_t mg_log, mg_axis, mg_len, mg_off;
  /* Tap-major form of the reduction axis (Kpad_t > 0; planner: every operand with >= 8 channels and <= 112 taps), used
   * by the bf16x3 / bf16 kernels: k runs over 8-channel groups g = t * gpt + c8 (tap t, channels 8*c8 .. 8*c8+7),
   * Kpad_t = 16 * ceil(ntaps * gpt / 2); the table passed as `ktab` then has ntaps + 16 rows (off = offset of channel 0
   * of the tap, da, db; 16 invalid tail rows) and In(n, k, a, b) adds channel * in_cs.  in_extent = bytes spanned by one
   * sample of the operand: reads beyond it (padded channels of the last group) return 0.  The planner may cut the channels into
   * blocks and present every (block, tap) as one table row (gpt = groups per block).  gpt2: groups per table row of the SECOND
   * phase's table in two-phase launches (0 = same as gpt). */
  int32_t Kpad_t, gpt, ntaps, gpt2;
  int64_t in_ns, in_as, in_bs;
  int64_t out_ns, out_cs, out_as, out_bs;
  int64_t in_cs, in_extent;
  /* bf16 STORAGE of single operands (bf16 arithmetic mode only; strides stay in elements, in_extent in bytes):
   * in_bf16: the gathered operand `in` (forward family) / the input operand x (rfx_gemm_wgrad) holds bf16 values -- tap-major
   *          tiled kernels only; 3 = channels-last bf16 operand (in_cs == 1,
   *          position strides multiples of 8): one 16-byte load per lane and K step -- probe of the next layout,
   *          not used by the product path;  out_bf16: `out` is written as bf16 (forward family, RNE; GroupNorm statistics of the epilogue
   *          are those of the rounded values) / the gradient operand g of rfx_gemm_wgrad holds bf16 values.
   * A tensor consumed only as a GEMM operand in bf16 mode loses nothing by being stored in 16 bits: the MFMA rounds it anyway. */
  int32_t in_bf16, out_bf16;
  /* Halo-tile form (halo_nt > 0; bf16 arithmetic mode, single-phase, unit input strides, OB % 128 == 0; planner:
   * convplan.halo_geometry): the table's first halo_nt rows are the REAL taps (3 or 9; the remaining rows repeat them per
   * channel block), whose displacements span rows [halo_da0, halo_da0 + halo_rows) and columns [halo_db0, halo_db0 + halo_w - 128]
   * around an output position.  gemm_halo_kernel (csrc/gemm_halo.h) stages that input tile once per 16-channel chunk in LDS and
   * serves every tap from it; same packed A, same results up to the order of the fp32 accumulation.  0 = tap-major kernels. */
  int32_t halo_nt, halo_rows, halo_w, halo_da0, halo_db0, halo_pad;
} rfx_gemm_desc;

/* Epilogue: v = acc + bias[m]; v = act(v); [second GEMM phase accumulates into
 * act(v)]; v += res[...] ; out = v.  All pointers may be NULL. */
typedef struct rfx_epilogue {
  const float* bias;      /* [M] */
  int32_t act;            /* enum rfx_act, applied after bias (after phase 1) */
  const float* act_param; /* PReLU slopes [M] */
  const float* res;       /* residual added last, indexed with out coordinates */
  int64_t res_ns, res_cs, res_as, res_bs;
  int32_t act2;           /* activation applied at the very end (after phase 2 / residual) */
  /* backward-of-activation mode (bwd != 0): the GEMM recomputes the pre-activation
   * v = acc + bias, `res` holds the incoming gradient G (out coordinates) and the
   * kernel writes out = G * act'(v); for PReLU, gparam[m] += sum G * min(v, 0).
   * Used to re-materialise PReLU(conv1(x)) in the TCN backward instead of
   * storing a second 256 x T activation per block. */
  int32_t bwd;
  float* gparam;
  /* optional: stat_sums[2n] += sum v, stat_sums[2n+1] += sum v^2 over the values stored for sample n (fp64;
   * caller zeroes): GroupNorm(1, C) statistics of the NEXT layer without re-reading the tensor */
  double* stat_sums;
  /* > 1: stat_sums is [N][stat_slots][2] and workgroups spread their atomics over the slots (power of two): a sample whose
   * positions span thousands of workgroups would otherwise serialise them on two addresses.  rfx_groupnorm_fwd takes the
   * same count in sums_given and adds the slots up. */
  int32_t stat_slots;     /* in bwd mode the same count applies to gparam: [stat_slots][M] partial sums */
  /* GLU fused into the store (HDemucs rewrite conv -> GLU, torchaudio HDemucs via models.py:319): the GEMM rows are the
   * conv's 2C output channels INTERLEAVED, row 2c = channel c ("a"), row 2c+1 = channel C+c ("b"); the kernel stores
   * the conv output in its natural channel order (needed by the backward) and glu_out[n*glu_ns + c*out_cs + pos] =
   * a * sigmoid(b).  bias is indexed in natural channel order.  Not combined with res / act2 / stat_sums / bwd / merged. */
  float* glu_out;
  int64_t glu_ns;
} rfx_epilogue;

/* Arithmetic of the MFMA gather-GEMM.  RFX_PREC_F32: v_mfma_f32_32x32x2_f32, exact fp32 products.
 * RFX_PREC_BF16X3: every fp32 operand is split into two bf16 (hi + lo) and the product is
 * hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- ~2^-16 relative error per
 * product, 3/16 of the fp32 MFMA cost.  The packed buffer has the same size in both modes. */
/* RFX_PREC_BF16: operands rounded to bf16 (RNE), one MFMA per product, fp32 accumulation -- the arithmetic of
 * `trainer.precision=bf16-mixed` (torch autocast rounds conv / linear operands the same way; storage, accumulators,
 * norms, FFT and losses stay fp32 here).  BF16X3 / BF16 forward launches need the tap-major form (Kpad_t > 0); the
 * planner falls back to RFX_PREC_F32 (exact) for operands with fewer than 8 channels. */
enum rfx_gemm_prec { RFX_PREC_F32 = 0, RFX_PREC_BF16X3 = 1, RFX_PREC_BF16 = 2 };

/* (woff[k] < 0: zero row -- channel padding of the tap-major order.)
 * A[k][m] = w[m*w_ms + woff[k]]  (k < K), zero padded to [Kpad + 64][Mpad]: the packed
 * matrix carries four extra all-zero K steps and every ktab passed to rfx_gemm_fwd
 * carries Kpad + 96 rows (the tail rows invalid: da = -2^30) so that the MFMA kernel's
 * operand prefetch is branch-free. */
int rfx_pack_a(const float* w, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K,
               int32_t Mpad, int32_t Kpad, int32_t prec, float* apack, void* stream);
/* w[m*w_ms + woff[k]] += sum over the `splits` slices [M][Kpad] rfx_gemm_wgrad left in dapack, added in a fixed order
 * (deterministic: the weight-gradient family carries no atomics). */
int rfx_unpack_add(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K,
                   int32_t Kpad, float* dw, int32_t splits, void* stream);
/* rfx_unpack_add + the bias gradient in the same launch: db[m] += sum over splits of dapack[split][m][bias_col] (the constant-one
 * column rfx_gemm_wgrad appends for plans built with a bias row). */
int rfx_unpack_add_bias(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K, int32_t Kpad, float* dw,
                        int32_t bias_col, float* db, int32_t splits, void* stream);
/* Same with `=` instead of `+=`: for a plan whose K rows cover every weight element exactly once (a dense
 * convolution's own plan) the caller need not zero-fill dw first. */
int rfx_unpack_set(const float* dapack, const int32_t* woff, int64_t w_ms, int32_t M, int32_t K,
                   int32_t Kpad, float* dw, int32_t splits, void* stream);
/* out[m] = sum over splits of dapack[split][m][col] (the bias-gradient column, for callers that return fresh tensors). */
int rfx_unpack_col(const float* dapack, int32_t M, int32_t Kpad, int32_t col, int32_t splits, float* out, void* stream);

/* Forward gather-GEMM on the fp32 MFMA path (v_mfma_f32_32x32x2_f32).
 * Optional second phase (apack2/ktab2/K2 != 0): after phase 1 the epilogue
 * activation is applied to the accumulators, then phase 2 keeps accumulating
 * (TCNBlock: PReLU(conv1(x)) + res(x), tcn.py:48-59, in one launch). */
int rfx_gemm_fwd(const rfx_gemm_desc* d, const float* apack, const rfx_ktab_entry* ktab,
                 const float* in, float* out, const rfx_epilogue* epi,
                 const float* apack2, const rfx_ktab_entry* ktab2, int32_t K2, int32_t Kpad2,
                 const float* in2 /* phase-2 input, same strides as `in`; NULL = `in` */,
                 int32_t prec /* enum rfx_gemm_prec; must match the packing; thin (M <= 8) path is always fp32 */,
                 void* stream);

/* Weight gradient of the same descriptor:
 * dapack[m][k] += sum_{n,a,b} g[n*g_ns + m*g_cs + a'*g_as + b'*g_bs] * In(n,k,a,b)   (dapack: [M][Kpad])
 * (g indexed with OUT coordinates/strides of the descriptor: out_* fields).
 * Rows k with ktab flag bit0 use In == 1 (bias gradient).  The positions are cut into *splits_out splits (chosen here, at most
 * ws_floats / (M * Kpad)); split s STORES its partial matrix to dapack + s * M * Kpad -- no zero fill, no atomics; the unpack
 * entry points add the slices in a fixed order. */
int rfx_gemm_wgrad(const rfx_gemm_desc* d, const rfx_ktab_entry* ktab, const float* in,
                   const float* g, float* dapack, int64_t ws_floats, int32_t* splits_out, int32_t prec, void* stream);

/* ---- framed FFT front / back end ---------------------------------------------
 * rfx_fft_analysis : frames -> window -> real FFT -> epilogue.  As STFT it reproduces
 *   torch.stft(center=True, pad_mode="reflect", onesided) : utils.py:148-154
 *   (spectrogram), HDemucs _spec, auraloss STFTLoss, Separator, MelSpectrogram.
 *   With in_mode=1 / herm=1 it is the backward of the iSTFT.
 * rfx_fft_synthesis: spectrum -> inverse real FFT -> window -> overlap-add BY OWNERSHIP
 *   (round 6: a workgroup walks consecutive frames of a row, carries the incomplete tail
 *   of the running sum in its own scratch and stores every output sample exactly once: no atomics,
 *   no zero fill, bit-reproducible; d->accum selects write / add).  Geometries outside
 *   that form (extra reflect pads, rows shorter than 2 n_fft in the reflect mode) fall
 *   back to fp32 atomics into an output the launcher zero-fills itself.  With herm=1 / in_mode=1 it is
 *   torch.istft (HDemucs _ispec, Separator); with herm=0 / in_mode=0 it is the
 *   backward (adjoint) of the STFT.
 * x: [R][T]; frame f covers padded samples [f*hop, f*hop + n_fft); the hann window
 * of `win` samples is centred in n_fft.  n_fft in {512, 1024, 2048, 4096}.
 */
enum rfx_stft_out {
  RFX_STFT_COMPLEX = 0, /* [R][bins][frames_out][2] (torch view_as_real layout) */
  RFX_STFT_CAC = 1,     /* [R][2][bins][frames_out]: real plane, imag plane (HDemucs _magnitude) */
  RFX_STFT_MAG = 2,     /* [R][bins][frames_out] = sqrt(max(re^2+im^2, eps)) (auraloss) */
  RFX_STFT_POW = 3,     /* re^2+im^2 (MelSpectrogram power=2) */
  RFX_STFT_MAGPOW = 4,  /* (sqrt(re^2+im^2) + eps) ^ alpha (utils.py:159) */
  RFX_STFT_COMPLEX_FM = 5 /* [R][frames_out][bins][2]: frame-major complex, for consumers that are layout-free (the MR-STFT loss
                             kernels, models.py:320): every frame's bins are one contiguous run, so the FFT kernels store / load
                             full lines instead of 8-64 byte pieces of a [bin][frame] transpose */
};
typedef struct rfx_stft_desc {
  int32_t R, T;           /* rows (batch*channels), samples per row of the time signal */
  int32_t n_fft, hop, win;
  int32_t bins;           /* bins stored: n_fft/2+1, or n_fft/2 to drop Nyquist (HDemucs) */
  int32_t frame0;         /* first frame stored (HDemucs keeps [2 : 2+le]) */
  int32_t frames_out;     /* number of frames stored */
  int32_t mode;           /* enum rfx_stft_out (synthesis: COMPLEX or CAC = spectrum layout) */
  int32_t extra_pad_l, extra_pad_r; /* in_mode 0: extra reflect padding applied first (HDemucs _spec) */
  int32_t in_mode;        /* 0: sample = reflect(p - n_fft/2); 1: sample = p - in_offset, zero outside */
  int32_t in_offset;
  int32_t herm;           /* 1: irfft semantics (istft fwd / bwd); 0: plain one-sided rfft adjoint */
  float scale;            /* multiplies every windowed sample */
  float eps, alpha;
  int32_t accum;          /* synthesis: 0 = every element of out is WRITTEN (no initialisation needed); 1 = out += (the second and
                             later resolutions of the MR-STFT loss gradient add into the first one's result) */
} rfx_stft_desc;

int rfx_fft_analysis(const rfx_stft_desc* d, const float* x, const float* window, const float* mul,
                     float* out, void* stream);
/* ws: rfx_fft_synthesis_ws(d) floats of scratch (no initialisation needed): the running sums a workgroup carries from one frame
 * batch to the next and the padded edge zones of the reflect-padded adjoint */
int64_t rfx_fft_synthesis_ws(const rfx_stft_desc* d);
int rfx_fft_synthesis(const rfx_stft_desc* d, const float* spec, const float* window, const float* mul,
                      float* ws, float* out, void* stream);

/* ---- LSTM recurrence ------------------------------------------------------------
 * Bidirectional nn.LSTM layer (HDemucs DConv BLSTM through models.py:319; Open-Unmix models.py:297-298), one
 * persistent launch per sweep.  All sequence tensors are channel-major [C][P], P = T*Bn, position p = t*Bn + b.
 * The caller runs the dense parts as gather-GEMMs: xp = [W_ih; W_ih_reverse] x + (b_ih + b_hh) before
 * rfx_lstm_fwd, and dX / dW_ih / dW_hh / db from dG after rfx_lstm_bwd.  Gate order i, f, g, o (PyTorch).
 * H % 32 == 0, H <= 512.  Arithmetic: bf16x3 split products, fp32 accumulate / state.
 * ws: caller-owned scratch of rfx_lstm_ws_bytes(H) bytes (cluster exchange buffers + arrival counters), reusable
 * across calls on one stream; its first int32 is a sticky error flag the caller zeroes once and may poll
 * (non-zero = a bounded spin timed out, results invalid). */
/* bytes of one direction's packed W_hh (MFMA fragments for both sweeps) */
int rfx_lstm_pack_bytes(int32_t H);
int rfx_lstm_ws_bytes(int32_t H);
/* whh: [4H][H] row-major -> pack (rfx_lstm_pack_bytes(H) bytes).  The two directions are packed back to back. */
int rfx_lstm_pack(const float* whh, int32_t H, void* pack, void* stream);
/* Host glue of one bidirectional nn.LSTM layer as single launches (models.py:319 / 297-298 reach it through torchaudio HDemucs'
 * _BLSTM and Open-Unmix): wcat [8H][Cin] = [w_ih ; w_ih_r] and bcat [8H] = [b_ih + b_hh ; b_ih_r + b_hh_r], the operands of the
 * input-projection GEMM of both directions ... */
int rfx_lstm_cat_params(const float* w_ih, const float* w_ih_r, const float* b_ih, const float* b_hh, const float* b_ih_r,
                        const float* b_hh_r, int32_t H, int32_t Cin, float* wcat, float* bcat, void* stream);
/* ... and that GEMM's weight / bias gradients (dwcat [8H][Cin], dbcat [8H]) ADDED into the six parameter gradients they belong to
 * (both biases of a direction receive the same gradient). */
int rfx_lstm_grad_scatter(const float* dwcat, const float* dbcat, int32_t H, int32_t Cin, float* g_w_ih, float* g_w_ih_r,
                          float* g_b_ih, float* g_b_hh, float* g_b_ih_r, float* g_b_hh_r, void* stream);
/* Single-workgroup form of the recurrence (H = 192, bf16 operands, small batches: one workgroup per (16-sequence tile, direction), W_hh
 * register-resident, exchange through LDS).  rfx_lstm_local: 1 if rfx_lstm_fwd (bwd = 0) / rfx_lstm_bwd (bwd = 1) will take it for this
 * shape and arithmetic mode; the caller must then also have called rfx_lstm_pack_local on the same `pack` buffer (per direction). */
int rfx_lstm_local(int32_t H, int32_t Bn, int32_t prec, int32_t bwd);
int rfx_lstm_pack_local(const float* whh, int32_t H, void* pack, void* stream);
/* xp [2][4H][P]; pack = both directions; out [2H][P] (forward dir rows 0..H-1, reverse H..2H-1);
 * gates [2][4H][P] and cstate [2][H][P] are saved for the backward sweep (both NULL for inference). */
/* prec: RFX_PREC_BF16 = h and W_hh rounded to bf16 (one MFMA per product: bf16-mixed); any other value = the bf16x3
 * split (fp32-grade products; also what the exact-fp32 GEMM mode uses for the recurrence). */
int rfx_lstm_fwd(const float* xp, const void* pack, int32_t T, int32_t Bn, int32_t H, float* out, float* gates,
                 float* cstate, void* ws, int32_t prec, void* stream);
/* gout [2H][P] -> dG [2][4H][P], the gradients of the gate pre-activations. */
int rfx_lstm_bwd(const float* gout, const void* pack, const float* gates, const float* cstate, int32_t T,
                 int32_t Bn, int32_t H, float* dG, void* ws, int32_t prec, void* stream);
/* Form choice of the H = 192 bf16 recurrence (no reference counterpart: torch.nn.LSTM has one form): launches of at most
 * fwd_max_sequences / bwd_max_sequences sequences take the single-workgroup form (defaults 64 / 16, RFX_LSTM_LOCAL /
 * RFX_LSTM_LOCAL_BWD), larger ones the wave-cluster form; 0 = always the cluster form, negative = back to the defaults.  Both
 * forms meet the oracle to the mode's tolerance but round h_t after differently ordered sums, so a test that compares a batch
 * with its single clips at 1e-6 pins one form. */
int rfx_lstm_set_local(int32_t fwd_max_sequences, int32_t bwd_max_sequences);

/* ---- elementwise / reductions ------------------------------------------------ */
/* y = act(x) elementwise over n contiguous floats; PReLU/bias not supported here. */
int rfx_act_fwd(const float* x, float* y, int64_t n, int32_t act, void* stream);
/* gx = gy * act'(x) */
int rfx_act_bwd(const float* x, const float* gy, float* gx, int64_t n, int32_t act, void* stream);
/* y = act(x) + res (same shapes, contiguous): decoder GELU + the next layer's skip-connection add (HDemucs, models.py:319) */
int rfx_act_add_fwd(const float* x, const float* res, float* y, int64_t n, int32_t act, void* stream);
/* The same between row layouts: x, gy (NULL: forward, out = act(x); else out = gy * act'(x)) and out are (D0, D1, D2)
 * grids of contiguous T-float rows with their own row strides (in floats).  Used where the reference permutes around
 * an activation -- torchaudio HDemucs _HEncLayer: `y = gelu(norm1(conv(x)))` then `y.permute(0, 2, 1, 3).reshape(-1, C, T)`
 * for the DConv branch (reached through remfx/models.py:319) -- so that the permuted copy is never made. */
int rfx_act_rows(const float* x, int64_t xs0, int64_t xs1, int64_t xs2, const float* gy, int64_t gs0, int64_t gs1,
                 int64_t gs2, float* out, int64_t os0, int64_t os1, int64_t os2, int32_t D0, int32_t D1, int32_t D2,
                 int32_t T, int32_t act, void* stream);
/* The same with x STORED as bf16 (a conv output of the bf16 arithmetic mode): gy == NULL: out (fp32) = act(x); otherwise out is the
 * bf16 gradient of that bf16 tensor (strides of x / of the backward output in bf16 elements).  Replaces the same call sites as
 * rfx_act_rows when trainer.precision = bf16-mixed keeps the encoder conv outputs in 16 bits (what torch autocast stores). */
int rfx_act_rows16(const void* x, int64_t xs0, int64_t xs1, int64_t xs2, const float* gy, int64_t gs0, int64_t gs1,
                   int64_t gs2, void* out, int64_t os0, int64_t os1, int64_t os2, int32_t D0, int32_t D1, int32_t D2,
                   int32_t T, int32_t act, void* stream);
/* y = a * b elementwise (contiguous, n elements; 16-byte aligned): mask * representation and tanh * sigmoid gating of asteroid's
 * DPTNet (reference remfx/models.py:327-344). */
int rfx_mul(const float* a, const float* b, float* y, int64_t n, void* stream);
/* nn.PReLU over [N][C][L] contiguous with a per-channel slope (C = 1: the single-parameter form of asteroid DPTNet's
 * `first_out`, reference remfx/models.py:327-344; per-channel: remfx/tcn.py:46).  Forward; backward: gx, and gslope[C] WRITTEN (one fp64
 * slot per (channel, sample) in `ws`, N * C doubles, added in sample order). */
int rfx_prelu_fwd(const float* x, const float* slope, float* y, int64_t N, int64_t C, int64_t L, void* stream);
int rfx_prelu_bwd(const float* x, const float* gy, const float* slope, float* gx, double* ws, float* gslope,
                  int64_t N, int64_t C, int64_t L, void* stream);

/* out[c] = sum over (n, a, b) of x[n*ns + c*cs + a*as + b*bs]  (bias gradients; written, not accumulated).  ws: rfx_channel_sum_ws(...)
 * doubles of per-workgroup partials (no initialisation needed), added in a fixed order: bit-reproducible. */
int64_t rfx_channel_sum_ws(const float* x, int32_t N, int32_t C, int32_t A, int32_t B, int64_t ns, int64_t cs, int64_t as, int64_t bs);
int rfx_channel_sum(const float* x, int32_t N, int32_t C, int32_t A, int32_t B, int64_t ns, int64_t cs,
                    int64_t as, int64_t bs, double* ws, float* out, void* stream);

/* out = x + alpha * y, x / out contiguous (N, C, A, B), y read through element strides (0 = broadcast):
 * skip / inject / frequency-embedding adds inside HDemucs (models.py:319). */
int rfx_add_bcast(const float* x, const float* y, float* out, int64_t N, int32_t C, int32_t A, int32_t B,
                  int64_t yn, int64_t yc, int64_t ya, int64_t yb, float alpha, void* stream);
/* per-row mean / unbiased std of x[R][L] and out = x*a[r] + b[r] (b may be NULL): HDemucs input and spectrogram standardisation /
 * de-standardisation (x.mean/std over dims 1..).  sums: fp64 workspace of 2 * R * rfx_row_moments_slots(L) values, no initialisation
 * needed (one slot per workgroup, added in slot order: bit-reproducible, no atomics).  coef_a / coef_b (both or neither): the
 * standardisation as one affine map, a = 1 / (eps + std), b = -mean * a. */
int rfx_row_moments_slots(int64_t L);
int rfx_row_moments(const float* x, int32_t R, int64_t L, double* sums, float* mean, float* stdv, float eps, float* coef_a,
                    float* coef_b, void* stream);
int rfx_row_affine(const float* x, const float* a, const float* b, float* out, int32_t R, int64_t L, void* stream);

/* Inverted dropout with a counter-based mask (keep(i) = u24(splitmix64(seed, i)) >= p; out = x / (1 - p) or 0); the backward pass
 * is the same call on the gradient with the same seed.  Replaces F.dropout(train=True) in Cnn14 (classifier.py:211-284) and the
 * inter-layer dropout of Open-Unmix's nn.LSTM (models.py:298; un-vendored open-unmix 1.2.1).  0 <= p < 1. */
int rfx_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream);

/* x[r][f][t] = 0 where f0[r] <= f < f1[r] or t0[r] <= t < t1[r], in place; x: (R, F, T) contiguous.
 * torchaudio FrequencyMasking / TimeMasking (iid masks) as Cnn14 applies them in training when specaugment is set:
 * classifier.py:185-187, 198-204.  Span bounds are R-length int32 device vectors drawn by the caller. */
int rfx_span_mask(float* x, int32_t R, int32_t F, int32_t T, const int32_t* f0, const int32_t* f1, const int32_t* t0,
                  const int32_t* t1, void* stream);

/* Overlapping-frame layout of the Hybrid Demucs BLSTM (torchaudio `_BLSTM`: sequences longer than 200 steps run as
 * frames of `width` 200 at `stride` 100, the central parts are stitched back; reached from models.py:319).
 * x: (B, C, T) contiguous; h: (C, width * Bn) with Bn = B * nfr, position t * Bn + b * nfr + k = step t of frame k of row b
 * (the channel-major layout of rfx_lstm_fwd).  mode 0: h = frames of x (zero beyond T);  1: x = adjoint of mode 0 applied to h;
 * 2: x = stitched central parts of h (+ skip[b][c][tau] when skip != NULL);  3: h = adjoint of mode 2 applied to x. */
int rfx_blstm_frames(const float* src, const float* skip, float* dst, int32_t B, int32_t C, int32_t T, int32_t nfr,
                     int32_t width, int32_t stride, int32_t mode, void* stream);

/* out[0] = scale * sum |a-b| over n elements (nn.L1Loss, models.py:320: scale = 1 / n).  ws: RFX_L1_SLOTS doubles of per-workgroup
 * partials (no initialisation needed), added in slot order. */
#define RFX_L1_SLOTS 512
int rfx_l1_sum(const float* a, const float* b, int64_t n, double* ws, float scale, float* out, void* stream);

/* ---- LocalState attention (torchaudio HDemucs `_LocalState` inside the DConv blocks, reached from models.py:319) ----
 * q, k, cont: (B, heads*ch, T) contiguous, channel = head*ch + c; qd: (B, heads*nd, T) raw decay projections.
 *   dots[t, s] = <k[:, t], q[:, s]> / sqrt(ch) - sum_f (f+1) |t-s| / sqrt(nd) * sigmoid(qd[f, s]) / 2;  dots[s, s] = -100
 *   w = softmax over t (stored to w (B, heads, T, T) when non-NULL);  out[c, s] = sum_t w[t, s] cont[c, t]
 * T <= 256, ch * T <= 12288, nd <= 8.  The backward returns the gradients of q, k, cont and the raw qd. */
int rfx_localstate_fwd(const float* q, const float* k, const float* cont, const float* qd, int32_t B, int32_t heads,
                       int32_t ch, int32_t T, int32_t nd, float* w, float* out, void* stream);
int rfx_localstate_bwd(const float* q, const float* k, const float* cont, const float* qd, const float* w,
                       const float* gout, int32_t B, int32_t heads, int32_t ch, int32_t T, int32_t nd, float* dq, float* dk,
                       float* dcont, float* dqd, void* stream);

/* The same operator on the bf16 matrix pipe (bf16 arithmetic mode = torch autocast: both products on bf16 operands, fp32
 * accumulation, fp32 softmax): flash-style, the (T, T) weights stay in registers and are recomputed in the backward pass, so
 * there is no w tensor.  ch in {16, 32, 48, 64, 96}, T <= 256, nd <= 8; rfx_localstate_mfma_ok returns 1 for supported shapes
 * (callers fall back to rfx_localstate_fwd / bwd otherwise). */
int rfx_localstate_mfma_ok(int32_t B, int32_t heads, int32_t ch, int32_t T, int32_t nd);
int rfx_localstate_mfma_fwd(const float* q, const float* k, const float* cont, const float* qd, int32_t B, int32_t heads,
                            int32_t ch, int32_t T, int32_t nd, float* out, void* stream);
int rfx_localstate_mfma_bwd(const float* q, const float* k, const float* cont, const float* qd, const float* gout, int32_t B,
                            int32_t heads, int32_t ch, int32_t T, int32_t nd, float* dq, float* dk, float* dcont, float* dqd,
                            float* stat /* workspace, B*heads*T*4 floats, 16-byte aligned */, void* stream);
/* Any-T path (whole files: more than 256 frames per clip; same operator, torchaudio HDemucs `_LocalState` reached from
 * remfx/models.py:319 / scripts/remfx_detect.py:44-61): keys streamed through LDS, O(T) memory, exact fp32.  stat: B*heads*T*4
 * floats {max, sum, <out, gout>, -}; the forward writes the first two, the backward reads them (and `out`) and fills the third.
 * ch <= 104, nd <= 64.  Returns -1 for unsupported shapes. */
int rfx_localstate_gen_fwd(const float* q, const float* k, const float* cont, const float* qd, int32_t B, int32_t heads,
                           int32_t ch, int32_t T, int32_t nd, float* stat, float* out, void* stream);
int rfx_localstate_gen_bwd(const float* q, const float* k, const float* cont, const float* qd, float* stat, const float* out,
                           const float* gout, int32_t B, int32_t heads, int32_t ch, int32_t T, int32_t nd, float* dq, float* dk,
                           float* dcont, float* dqd, void* stream);

/* ---- audio-effect rendering on the device ---------------------------------------------------------------
 * The reference renders its training effects on the CPU with pedalboard (JUCE DSP) and normalises loudness with pyloudnorm
 * after every effect: remfx/effects.py:297-616 (RandomPedalboard{Distortion, Delay, Chorus, Compressor, Reverb}.forward),
 * :619-629 (LoudnessNormalize), called from remfx/datasets.py:109-202 (parallel_process_effects) and :205-330
 * (DynamicEffectDataset.process_effects).  x, y: (B, T) fp32 contiguous mono clips; every parameter is a B-vector on the device
 * (each clip draws its own).  y may alias x except for rfx_fx_delay.  Algorithms: csrc/fx.hip; oracle/ref_effects.py. */
int rfx_fx_distortion(const float* x, float* y, int32_t B, int64_t T, const float* gain /* 10^(drive_db / 20) */, void* stream);
int rfx_fx_delay(const float* x, float* y, int32_t B, int64_t T, const int32_t* delay_samples, const float* feedback,
                 const float* mix, void* stream);
int rfx_fx_chorus(const float* x, float* y, int32_t B, int64_t T, float sample_rate, const float* rate_hz, const float* depth,
                  const float* centre_delay_ms, const float* feedback, const float* mix, void* stream);
int rfx_fx_compressor(const float* x, float* y, float* env_ws /* (B, T) workspace */, int32_t B, int64_t T,
                      const float* threshold_lin, const float* ratio, const float* c_attack, const float* c_release, void* stream);
int rfx_fx_reverb(const float* x, float* y, int32_t B, int64_t T, int32_t sample_rate, const float* damp, const float* feedback,
                  const float* wet1, const float* dry, void* stream);
/* BS.1770 integrated loudness (pyloudnorm.Meter.integrated_loudness, K-weighting, 400 ms blocks at 75 % overlap, absolute and
 * relative gates) and the LoudnessNormalize gain 10^(clamp(target - L, -120, 40) / 20) per clip.  coef: 28 HOST doubles = the two
 * normalised biquads b1[3] a1[3] b2[3] a2[3] and the 4 x 4 zero-input state transition of `chunk` samples (row-major);
 * chunk * 64 >= T; hop_len = samples per 100 ms; nblk gating blocks over nhop >= nblk + 3 hops; inv_block = 1 / (0.4 * rate);
 * hop_ws: B * nhop doubles.  Outputs lufs[B], gain[B]; apply the gain with rfx_fx_scale. */
int rfx_fx_loudness(const float* x, int32_t B, int64_t T, int32_t chunk, int32_t hop_len, int32_t nhop, int32_t nblk,
                    double inv_block, const double* coef, float target_lufs, double* hop_ws, float* lufs, float* gain, void* stream);
int rfx_fx_scale(const float* x, float* y, int32_t B, int64_t T, const float* gain, void* stream);

/* ---- fused DConv depth-layer of the Hybrid Demucs frequency branch (bf16 arithmetic) ---------------------------------------
 * torchaudio HDemucs `_DConv` layer (reached from remfx/models.py:319): x_out = x + scale * GLU(GN(conv1x1(GELU(GN(conv3_dil(x))))))
 * for N samples of (C = 48, T = 256), hidden = C / 4 = 12, dilation 1 or 2, GroupNorm(1, .) with `eps`: ONE launch, one pass over x
 * (csrc/dconv.hip).  Weights in PyTorch layout: w1 (12, 48, 3), w2 (96, 12[, 1]).  rfx_dconv_layer_ok: shape test.
 * Training: pass all four of h16 (N, 12, T) bf16 = conv1 output, z16 (N, 96, T) bf16 = conv2 output, a_out (N, 12, T) fp32 =
 * GELU(GN(h)), stats (4, N) fp32 = mean1, rstd1, mean2, rstd2 -- exactly what rfx_groupnorm_bwd_x16 / the gather-GEMM input- and
 * weight-gradient kernels of the layer-by-layer path read; inference: all four NULL. */
int rfx_dconv_layer_ok(int32_t C, int32_t T, int32_t dil);
int rfx_dconv_layer_fwd(const float* x, float* out, int32_t N, int32_t C, int32_t T, int32_t dil, const float* w1, const float* b1,
                        const float* gn1w, const float* gn1b, const float* w2, const float* b2, const float* gn2w,
                        const float* gn2b, const float* scale, float eps, void* h16, void* z16, float* a_out, float* stats,
                        void* stream);
/* Backward of the same layer in ONE launch (round 4): recomputes the forward from x, returns gx = dL/dx, the operands of the two
 * weight-gradient GEMMs -- dz (N, 2C, T) bf16 = gradient of the 1x1 convolution's output, a_out (N, H, T) fp32 = its input,
 * dh (N, H, T) bf16 = gradient of the 3-tap convolution's output (rfx_gemm_wgrad: dW2 = dz a^T, dW1 = dh * x) -- and `partial`:
 * rfx_dconv_layer_bwd_rows(N) rows of 5C + 2H floats [dscale C | dgn2w 2C | dgn2b 2C | dgn1w H | dgn1b H] whose column sums are the
 * LayerScale / GroupNorm gradients (one row per workgroup; the caller adds them up).  Replaces the autograd backward of torchaudio
 * `_DConv` reached from remfx/models.py:319. */
int rfx_dconv_layer_bwd_rows(int32_t N);
int rfx_dconv_layer_bwd(const float* x, const float* g, float* gx, int32_t N, int32_t C, int32_t T, int32_t dil, const float* w1,
                        const float* b1, const float* gn1w, const float* gn1b, const float* w2, const float* b2, const float* gn2w,
                        const float* gn2b, const float* scale, float eps, void* dz_bf16, float* a_out, void* dh_bf16, float* partial,
                        void* stream);

/* Label of the kernel instantiation rfx_gemm_fwd would launch (measurement only; see csrc/gemm.hip). */
int rfx_gemm_fwd_variant(const rfx_gemm_desc* d, const rfx_epilogue* epi, int32_t two_phase, int32_t prec);

/* Plain multi-head scaled-dot-product attention, out[c, s] = sum_t softmax_t(<k[:, t], q[:, s]> / sqrt(ch)) v[c, t], on the streaming
 * any-T kernels above; q, k, v: (B, heads * ch, T) channel-major.  The nn.MultiheadAttention core of asteroid's DPTNet
 * (`ImprovedTransformedLayer`; reference remfx/models.py:327-344, cfg/model/dptnet.yaml).  stat: B * heads * T * 4 floats. */
int rfx_mha_fwd(const float* q, const float* k, const float* v, int32_t B, int32_t heads, int32_t ch, int32_t T, float* stat,
                float* out, void* stream);
int rfx_mha_bwd(const float* q, const float* k, const float* v, float* stat, const float* out, const float* gout, int32_t B,
                int32_t heads, int32_t ch, int32_t T, float* dq, float* dk, float* dv, void* stream);

/* ---- GroupNorm (+ fused activation) --------------------------------------------
 * x: (N, C, S) contiguous, G groups.  mode: 0 y = gn(x); 1 y = gelu(gn(x));
 * 2 y = glu(gn(x)) -> (N, C/2, S); 3 y = res + scale[c] * glu(gn(x))  (DConv tail).
 * Replaces nn.GroupNorm + F.gelu / F.glu / _LayerScale inside torchaudio HDemucs
 * (models.py:319).  mean / rstd (N*G each) are written for the backward. */
int rfx_groupnorm_fwd(const float* x, const float* gamma, const float* beta, int32_t N, int32_t C, int32_t S,
                      int32_t G, float eps, int32_t mode, const float* res, const float* scale,
                      double* sums /* N*G*2 fp64 workspace; sums_given != 0: already holds {sum, sumsq} per
                                      (n, g) -- e.g. from rfx_epilogue.stat_sums -- and the statistics pass is skipped;
                                      sums_given = k > 1: the buffer is [N*G][k][2] partial sums (rfx_epilogue.stat_slots);
                                      sums_given = -1: see rfx_groupnorm_stat_chunks */,
                      int32_t sums_given, float* mean, float* rstd, float* y, void* stream);
/* Chunks the statistics kernel of rfx_groupnorm_fwd / _x16 cuts one group into.  sums_given = -1 in those entry points: no statistics
 * are given and `sums` is a workspace of N * G * chunks pairs of doubles that the kernel fills with plain stores, one pair per
 * (group, chunk), summed in chunk order -- no zero fill, no atomics (a fill followed by fp64 atomics lost contributions whenever a
 * second stream kept the machine busy).  sums_given = 0 keeps the N * G pair buffer + atomics form. */
int rfx_groupnorm_stat_chunks(int32_t C, int32_t S, int32_t G);
/* Floats of `work` the three backward entry points below need (G = 0: rfx_batchnorm_bwd): N*C*2 + N*(C/2) + 2 max(N*G, C) and, for
 * rows longer than one 4096-sample chunk, one slot per (sample, channel, chunk) partial -- plain stores added in chunk order (no zero
 * fill, no atomics: DESIGN.md 4.11). */
int64_t rfx_norm_bwd_work_floats(int32_t N, int32_t C, int32_t S, int32_t G);
/* dx (N, C, S); dgamma / dbeta (C) and dscale (C/2, mode 3) are OVERWRITTEN; `work` is a
 * caller-owned scratch of rfx_norm_bwd_work_floats(N, C, S, G) floats; the residual gradient of mode 3 is gy. */
int rfx_groupnorm_bwd(const float* x, const float* gamma, const float* beta, const float* mean,
                      const float* rstd, const float* gy, int32_t N, int32_t C, int32_t S, int32_t G,
                      int32_t mode, const float* scale, float* work, float* dx, float* dgamma, float* dbeta,
                      float* dscale, void* stream);

/* The same two calls with x -- and, in the backward pass, dx -- STORED as bf16 (a conv output of the bf16 arithmetic mode that
 * only this norm and GEMM operands ever read); y, res, gy stay fp32.  S % 4 == 0. */
int rfx_groupnorm_fwd_x16(const void* x, const float* gamma, const float* beta, int32_t N, int32_t C, int32_t S, int32_t G,
                          float eps, int32_t mode, const float* res, const float* scale, double* sums, int32_t sums_given,
                          float* mean, float* rstd, float* y, void* stream);
int rfx_groupnorm_bwd_x16(const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                          const float* gy, int32_t N, int32_t C, int32_t S, int32_t G, int32_t mode, const float* scale,
                          float* work, void* dx, float* dgamma, float* dbeta, float* dscale, void* stream);
/* BatchNorm2d (+ReLU, mode 4) over (N, S) per channel -- classifier.py:271-272 (ConvBlock).
 * use_given_stats != 0: eval mode, mean = running_mean, rstd = 1/sqrt(running_var + eps) are inputs;
 * otherwise batch statistics are computed into mean / rstd (biased variance).  sums: fp64 workspace of
 * C * rfx_batchnorm_stat_slots(N, S) PAIRS -- one per (channel, sample, 4096-value chunk), stored by the wave that owns it and added
 * in slot order (no zero fill, no atomics). */
int rfx_batchnorm_stat_slots(int32_t N, int32_t S);
int rfx_batchnorm_fwd(const float* x, const float* gamma, const float* beta, int32_t N, int32_t C, int32_t S,
                      float eps, int32_t mode, int32_t use_given_stats, double* sums, float* mean, float* rstd,
                      float* y, void* stream);
/* train-mode backward; work: rfx_norm_bwd_work_floats(N, C, S, 0) floats */
int rfx_batchnorm_bwd(const float* x, const float* gamma, const float* beta, const float* mean,
                      const float* rstd, const float* gy, int32_t N, int32_t C, int32_t S, int32_t mode,
                      float* work, float* dx, float* dgamma, float* dbeta, void* stream);
/* F.avg_pool2d(kernel = stride = (kh, kw)) on (NC, H, W) planes -- classifier.py:275 */
int rfx_avgpool2d_fwd(const float* x, float* y, int64_t NC, int32_t H, int32_t W, int32_t kh, int32_t kw,
                      void* stream);
int rfx_avgpool2d_bwd(const float* gy, float* gx, int64_t NC, int32_t H, int32_t W, int32_t kh, int32_t kw,
                      void* stream);
/* GLU over the channel axis of (N, C, S): y = x[:, :C/2] * sigmoid(x[:, C/2:]) */
int rfx_glu_fwd(const float* x, float* y, int64_t N, int64_t C, int64_t S, void* stream);
int rfx_glu_bwd(const float* x, const float* gy, float* gx, int64_t N, int64_t C, int64_t S, void* stream);
/* same with x (the conv output) and gx (its gradient) STORED as bf16 (bf16 arithmetic mode: both are only GEMM operands / GLU
 * inputs); gy fp32; (C/2)*S a multiple of 4. */
int rfx_glu_bwd_bf16(const void* x, const float* gy, void* gx, int64_t N, int64_t C, int64_t S, void* stream);

/* ---- complex-valued pieces of DCUNet (asteroid DCUNet via models.py:347-367) -------
 * Complex tensors are real tensors (N, 2C, S): channels [0,C) real parts, [C,2C) imaginary parts; a complex
 * convolution is then ONE rfx_gemm_fwd with the block weight [[Wr,-Wi],[Wi,Wr]].
 * sums (5, per channel c at c*5+q): sum xr, xi, xr^2, xr*xi, xi^2 in fp64 (ComplexBatchNorm statistics).
 * ws: fp64 workspace of 5 * C * rfx_cplx_slots(N, S) values (rfx_cplx_affine_act_bwd: 6 * C * rfx_cplx_slots) -- one slot per
 * (sample, 4096-value chunk), stored by the wave that owns it and added in slot order: no zero fill, no atomics. */
int64_t rfx_cplx_slots(int32_t N, int64_t S);
int rfx_cplx_moments(const float* x, int32_t N, int32_t C, int64_t S, double* ws, double* sums, void* stream);
/* ComplexBatchNorm coefficients (asteroid complex_nn BatchNorm inside DCUNet, models.py:356-367) in one launch: per channel
 * coef (6, C) = Zrr, Zri, Zir, Zii, Br', Bi' with y = Z x + B', Z = W V^{-1/2} (2x2 inverse square root of the covariance + eps)
 * and the mean folded into the bias.  Statistics come from `sums` (rfx_cplx_moments, inv_count = 1 / (N S); training) or from
 * stats_in (5, C) = running Mr, Mi, Vrr, Vri, Vii (eval) -- exactly one of the two.  stats_out (5, C), optional: the batch
 * statistics; RM* / RV*, optional (all five or none): running buffers updated in place, buf += momentum * (stat - buf).
 * rfx_cplx_coef_bwd: gcoef (6, C) -> gw (5, C) = gradients of Wrr, Wri, Wii, Br, Bi and, with sums given, cm (5, C) = the
 * gradient with respect to the five raw moments times inv_count (the `coef` operand of rfx_cplx_moments_bwd); cm may be NULL. */
int rfx_cplx_coef_fwd(const double* sums, double inv_count, const float* stats_in, const float* Wrr, const float* Wri,
                      const float* Wii, const float* Br, const float* Bi, float eps, int32_t C, float* coef, float* stats_out,
                      float* RMr, float* RMi, float* RVrr, float* RVri, float* RVii, float momentum, void* stream);
int rfx_cplx_coef_bwd(const double* sums, double inv_count, const float* stats_in, const float* Wrr, const float* Wri,
                      const float* Wii, const float* Br, const float* Bi, float eps, int32_t C, const float* gcoef, float* gw,
                      float* cm, void* stream);
/* gx += d(sum_q coef[q][c] * moment_q)/dx, coef: (5, C) fp32 */
int rfx_cplx_moments_bwd(const float* x, const float* coef, int32_t N, int32_t C, int64_t S, float* gx, void* stream);
/* out = leaky_relu(Z x + b, slope) per complex channel; coef (6, C) = Zrr, Zri, Zir, Zii, Br, Bi.
 * out[n*out_ns + c*S + s] (real) and out[n*out_ns + (c + out_im_off)*S + s] (imag): may be a slice of a
 * pre-allocated skip-concatenation buffer (no torch.cat copy). */
int rfx_cplx_affine_act_fwd(const float* x, const float* coef, int32_t N, int32_t C, int64_t S, float slope,
                            float* out, int64_t out_ns, int64_t out_im_off, void* stream);
/* gx (N, 2C, S) written, gcoef (6, C) overwritten */
int rfx_cplx_affine_act_bwd(const float* x, const float* coef, const float* gy, int64_t gy_ns, int64_t gy_im_off,
                            int32_t N, int32_t C, int64_t S, float slope, float* gx, double* ws, float* gcoef, void* stream);
/* BoundComplexMask("tanh") applied to the mixture STFT: out = tanh(|m|) m/|m| (*) tf; planes (N, 2, P) */
int rfx_bound_mask_fwd(const float* m, const float* tf, float* out, int32_t N, int64_t P, int64_t m_ns,
                       int64_t tf_ns, int64_t out_ns, void* stream);
int rfx_bound_mask_bwd(const float* m, const float* tf, const float* gout, float* gm, int32_t N, int64_t P,
                       int64_t m_ns, int64_t tf_ns, int64_t g_ns, int64_t gm_ns, void* stream);

/* Open-Unmix Separator (niter=0): out = mag * xc/|xc| over n complex bins (models.py:298,304) */
int rfx_phase_mask_fwd(const float* mag, const float* xc, float* out, int64_t n, void* stream);
int rfx_phase_mask_bwd(const float* xc, const float* gout, float* gmag, int64_t n, void* stream);

/* ---- losses -------------------------------------------------------------------
 * auraloss STFTLoss terms on complex spectra [R][n] (n = bins*frames, view_as_real layout):
 * sums[r] = { sum (|Y|-|X|)^2, sum |Y|^2, sum |log|X| - log|Y|| }, |.| = sqrt(max(re^2+im^2, eps)) (written, not accumulated).
 * Replaces auraloss.freq.STFTLoss.forward (models.py:299,320,362,385,107; metrics 237-255). */
#define RFX_STFT_REDUCE_SLOTS 64      /* ws: 3 * R * RFX_STFT_REDUCE_SLOTS doubles (per-workgroup partials, no initialisation needed) */
int rfx_stft_loss_reduce(const float* xc, const float* yc, int32_t R, int64_t n, float eps, double* ws, float* sums,
                         void* stream);
/* gxc = w_sc * d[sqrt(A_r)/sqrt(B_r)]/dxc + w_lm * d[sum |log|X|-log|Y||]/dxc  (A_r, B_r from sums).  gup (may be NULL): one
 * device float that multiplies both weights -- the upstream gradient of the scalar loss, read on the device so that the backward
 * pass needs no host synchronisation (and a training step can be captured into a hipGraph). */
int rfx_stft_loss_grad(const float* xc, const float* yc, int32_t R, int64_t n, float eps,
                       const float* sums, float w_sc, float w_lm, const float* gup, float* gxc, void* stream);
/* The forward of one STFTLoss resolution in ONE launch (replaces two rfx_fft_analysis + rfx_stft_loss_reduce; auraloss STFTLoss behind
 * models.py:320): two frames of a signal come out of one complex FFT (w s_a + i w s_b), the prediction's clamped powers wait in
 * registers for the target's and the three row sums are written to sums [R][3] (every workgroup stores its partial into its own slot of
 * ws -- rfx_stft_pair_loss_ws(d) doubles, no initialisation needed -- and the slots are added in order: bit-reproducible).  d: R, T, n_fft (512 / 1024 / 2048), hop, win, frames_out = 1 + T / hop,
 * bins = n_fft / 2 + 1, frame0 = 0, in_mode 0.  xspec / ymag (both or neither): the prediction's spectrum, frame-major complex
 * [R][frames][bins][2], and the clamped target magnitudes [R][frames][bins], for rfx_stft_loss_grad_m + rfx_fft_synthesis
 * (RFX_STFT_COMPLEX_FM) in the backward. */
int64_t rfx_stft_pair_loss_ws(const rfx_stft_desc* d);
int rfx_stft_pair_loss(const rfx_stft_desc* d, const float* x, const float* y, const float* window, float eps, double* ws, float* sums,
                       float* xspec, float* ymag, void* stream);
/* rfx_stft_loss_grad with the target given by its clamped magnitudes (rfx_stft_pair_loss's ymag) instead of its spectrum */
int rfx_stft_loss_grad_m(const float* xc, const float* ymag, int32_t R, int64_t n, float eps, const float* sums, float w_sc,
                         float w_lm, const float* gup, float* gxc, void* stream);
/* rfx_stft_loss_grad_m followed by rfx_fft_synthesis (in_mode 0, herm 0, RFX_STFT_COMPLEX_FM) in ONE launch: the gradient spectrum
 * is computed where the inverse transform's merge step loads it (same arithmetic, instruction for instruction) and never written --
 * 1.15 GB of the 64-clip step's traffic per resolution.  d as for rfx_fft_synthesis (d->accum: write / add).
 * The backward of auraloss STFTLoss behind models.py:320. */
int rfx_fft_synthesis_lossgrad(const rfx_stft_desc* d, const float* xspec, const float* ymag, const float* sums, float w_sc,
                               float w_lm, float eps, const float* gup, const float* window, float* ws, float* out, void* stream);
/* g[i] = w * gup[0] * sign(a[i] - b[i])   (nn.L1Loss backward; gup as above, NULL = 1) */
int rfx_l1_grad(const float* a, const float* b, int64_t n, float w, const float* gup, float* g, void* stream);
/* per row: sums[r] = { sum x, sum t, sum x t, sum x^2, sum t^2 } in fp64 (auraloss SISDRLoss).  ws: 5 * R * RFX_SISDR_SLOTS doubles of
 * per-workgroup partials (no initialisation needed), added in slot order. */
#define RFX_SISDR_SLOTS 128
int rfx_sisdr_sums(const float* x, const float* t, int32_t R, int64_t L, int64_t x_rs, int64_t t_rs,
                   double* ws, double* sums, void* stream);
/* The scalar tail of SISDRLoss in one launch: out[0] = -mean_r 10 log10(|a t|^2 / (|x - a t|^2 + eps) + eps), a = <x,t> / (|t|^2 + eps),
 * from the row sums of rfx_sisdr_sums (row means removed when zero_mean); auraloss SISDRLoss behind models.py:227-255. */
int rfx_sisdr_finish(const double* sums, int32_t R, int64_t L, int32_t zero_mean, double eps, float* out, void* stream);
/* The scalar tail of MultiResolutionSTFTLoss in one launch: sums[k] = the [R][3] row sums of resolution k (rfx_stft_pair_loss /
 * rfx_stft_loss_reduce), n[k] = spectrum cells per row; out[0] = mean_k (sc_k + lm_k) with sc_k = mean_r sqrt(A_r / B_r)
 * (per_example_sc) or sqrt(sum A / sum B), lm_k = sum_r C_r / (R n_k).  sums / n are HOST arrays of nres <= 8 entries.
 * auraloss MultiResolutionSTFTLoss behind models.py:320. */
int rfx_mrstft_combine(const float* const* sums, const int64_t* n, int32_t nres, int32_t R, int32_t per_example_sc, float* out,
                       void* stream);

/* ---- optimiser (flat fp32 buffers) ----------------------------------------------
 * Replaces torch.optim.AdamW.step + Lightning gradient_clip_val (models.py:185-191,
 * cfg/config.yaml:119). */
/* nbytes (multiple of 4) of zeros at p: optimizer.zero_grad() on the flat gradient buffer and the accumulation targets of the
 * atomics-based kernels (torch.zeros call sites of the hot path; Lightning's zero_grad behind models.py:185-191) */
int rfx_zero(void* p, int64_t nbytes, void* stream);
/* out[0] = sum g^2 (fp64).  ws: RFX_SUMSQ_SLOTS doubles of per-workgroup partials (no initialisation needed), added in slot order */
#define RFX_SUMSQ_SLOTS 4096
int rfx_sumsq(const float* g, int64_t n, double* ws, double* out, void* stream);
/* *coef = min(1, max_norm / (sqrt(*sumsq) * pre + 1e-6)) * pre ; *norm_out = sqrt(*sumsq) * pre */
int rfx_clip_coef(const double* sumsq, float max_norm, float pre, float* coef, float* norm_out, void* stream);
/* decoupled-weight-decay Adam, torch.optim.AdamW semantics; gradients are multiplied by
 * *gscale (device scalar, may be NULL) before use. */
int rfx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                   float eps, float wd, int32_t step, const float* gscale, void* stream);

/* ---- channels-last bf16 family (round 5; bf16 arithmetic mode = BASELINE config 3, trainer.precision=bf16-mixed) --------------
 * The frequency branch of Hybrid Demucs (torchaudio HDemucs freq_encoder / freq_decoder reached from remfx/models.py:308,317)
 * keeps its activations as [N][A][B][C] bf16 with the channels of one position contiguous: an MFMA B fragment is then one
 * 16-byte group, so operands go global -> LDS by DMA (`buffer_load ... lds`) with no per-element work and results leave through
 * an LDS transpose as full lines.  Every kernel of the family reads / writes such tensors; strides are in ELEMENTS. */
typedef struct rfx_cl_tensor {
  void* p;               /* bf16 [N][A][B][bs]; NULL = absent */
  int64_t ns, as;        /* element strides of n and of a row (a); a position (b) is bs elements */
  int32_t bs, c0;        /* channels stored per position; first channel this operand addresses */
} rfx_cl_tensor;

enum rfx_cl_epi {
  RFX_CL_STORE = 0,      /* out0 = v                                   (v = acc + bias [+ res]) */
  RFX_CL_GELU = 1,       /* out0 = v (may be absent), out1 = gelu(v) [+ aux0]   (encoder conv; decoder conv_tr + next skip) */
  RFX_CL_GLU = 2,        /* rows interleaved (a_c, b_c): out0 = v in natural order [a | b] (may be absent), out1 = a * sigmoid(b) */
  RFX_CL_DGELU = 3,      /* out0 = v (may be absent), out1 = v * gelu'(aux0)    (backward of GELU_ADD: skip gradient + pre-activation gradient) */
  RFX_CL_STORE_CM = 5,   /* v as fp32 CHANNEL-MAJOR through cm_out (rows = (sub-row | sub-position, channel), Co channels; BM = 32): the last
                          * decoder layer's transposed convolution, whose 1 - 2 output channels are no channels-last tensor */
  RFX_CL_DGLU = 4        /* aux0 = stored [a | b] of the forward GLU: out0 = [v * sigmoid(b) | v * a * sigmoid(b) (1 - sigmoid(b))]; out1 = v (optional) */
};

/* Implicit-GEMM convolution on channels-last operands (forward of Conv2d / ConvTranspose2d and their input gradients):
 *   acc[m](n, oa, b) = sum over row taps r, chunks c, column taps t, channels k of the chunk
 *                      W[m][r][c][t][k] * in[n][oa*SA + da[r]][b + db[t]][16 KS c + k]             (0 outside the tensor)
 * A workgroup (8 waves) owns BM rows x 256 consecutive b of one (n, oa); the reduction is cut into UNITS (r, c) of NTC column taps x
 * KS K steps of 16 channels whose B slab (256 [+ halo] positions x 16 KS channels) and packed A block arrive by LDS DMA through a
 * 2- or 3-deep ring, one workgroup barrier per unit.  OB == IB, OB % 256 == 0; row taps whose input row falls outside [0, IA) are skipped.
 * wrapb != 0: the B axis continues into the neighbouring row (1-D signals stored as [A][256]): halo columns come from linear memory
 * and only the sample's ends read as zero.
 * Merged phases (G > 1): GEMM row m = psi * Co + co is stored at output row oa * G + psi + g_off (dropped outside [0, OAo)),
 * channel co -- ConvTranspose2d with stride G along A, and the input gradient of a stride-G Conv2d, as ONE GEMM over 2 row taps. */
typedef struct rfx_cl_conv_desc {
  rfx_cl_tensor in;
  int32_t N, IA, IB, OA, OB, SA;
  int32_t NTR, NCH, NTC, KS;   /* row taps, channel chunks of 16 KS channels (coff = 16 KS c), column taps, K steps per chunk and tap */
  int32_t da0, da_step;   /* da[r] = da0 + r * da_step */
  int32_t db0, db_step;   /* db[t] = db0 + t * db_step, |db| <= 8 */
  int32_t wrapb;
  const void* apack;      /* rfx_cl_pack output: [NTR * NCH][MG][NTC][KS][BM / 32] 1-KiB MFMA A fragments */
  int32_t M, BM;          /* GEMM rows; rows per workgroup: 32, 64, 96 or 192 */
  int32_t mode;           /* enum rfx_cl_epi */
  int32_t G, g_off, OAo, Co;
  const float* bias;      /* the layer's bias in ITS channel order (GLU: [a | b]; merged / folded rows (Co < M): bias[m % Co]), or NULL */
  const float* rowadd;    /* RFX_CL_GLU only: fp32 [OA][M / 2] added to out1 (a per-row vector: the frequency embedding), or NULL */
  rfx_cl_tensor out0, out1, aux0, res;
  void* cm_out;           /* RFX_CL_STORE_CM: fp32 [N][Co][rows][positions], element strides cm_ns, cm_cs, cm_as */
  int64_t cm_ns, cm_cs, cm_as;
  int32_t cm_fold;        /* 0: merged row phases (output row oa*G + psi + g_off); 1: folded positions (position b * (M / Co) + psi) */
} rfx_cl_conv_desc;
int rfx_cl_conv(const rfx_cl_conv_desc* d, void* stream);
/* dst[i] = bf16(idx[i] < 0 ? 0 : src[idx[i]]): weights -> packed MFMA fragments (idx built once per layer by the host planner) */
int rfx_cl_pack(const float* src, const int32_t* idx, int64_t n, void* dst, void* stream);
/* channel-major fp32 / bf16 (N, C, A, B) [strides in elements, B contiguous] <-> channels-last bf16; B % 64 == 0, C % 8 == 0.
 * rfx_cl_from_cm fuses what would follow the conversion: v = src (+ res); mode 0: dst = v; 1: dst = v * gelu'(aux);
 * 2: GLU backward against aux = stored [a | b] (2 C channels), dst = [v * sigmoid(b) | v * a * sigmoid(b) (1 - sigmoid(b))];
 * 3: dst = gelu(v).  rfx_cl_to_cm with aux16 (bf16, channel-major, dst's strides; may be NULL): dst = v * gelu'(aux16). */
int rfx_cl_from_cm(const void* src, int32_t src_bf16, int64_t s_ns, int64_t s_cs, int64_t s_as, int32_t N, int32_t C, int32_t A,
                   int32_t B, const rfx_cl_tensor* dst, const rfx_cl_tensor* res, const rfx_cl_tensor* aux, int32_t mode, void* stream);
int rfx_cl_to_cm(const rfx_cl_tensor* src, int32_t N, int32_t C, int32_t A, int32_t B, void* dst, int32_t dst_bf16, int64_t d_ns,
                 int64_t d_cs, int64_t d_as, const void* aux16, void* stream);
/* dst [N][OA][B][16] bf16 = the 8 taps x Cs (1 | 2) channels of a (8, stride 4, padding 2) convolution gathered per output position from
 * the channel-major fp32 src [N][Cs][rows][positions] (strides in elements; along_b = 0: taps over rows 4 oa + k - 2, 1: over positions
 * 4 b + k - 2), channel k * Cs + c, zero outside and beyond 8 Cs: the operand of the 16-channel GEMMs that stand for the network's first
 * convolution (2 spectrogram / 1 waveform channels) and for the input gradient of its last transposed one. */
int rfx_cl_im2col_s4(const float* src, int64_t s_ns, int64_t s_cs, int64_t s_as, int32_t N, int32_t Cs, int32_t IA, int32_t IB, int32_t OA,
                     int32_t OB, int32_t along_b, void* dst, void* stream);
/* Frame-major ends of the Hybrid Demucs frequency branch (round 6; torchaudio HDemucs `_spec` / `_magnitude` / standardisation in front of
 * `freq_encoder[0]`, and the de-standardisation / `_mask` / `_ispec` behind `freq_decoder[-1]`, reached from models.py:319):
 * rfx_cl_im2col_fm: the 16-channel im2col operand of the first convolution from a FRAME-MAJOR spectrum src [N][F][bins][2] fp32
 *   (RFX_STFT_COMPLEX_FM) with the per-clip standardisation folded in: dst [N][bins / 4][F][16] bf16, element k * 2 + c of
 *   (n, oa, f) = a[n] src[n][f][4 oa + k - 2][c] + b[n], zero outside [0, bins).  bins % 4 == 0, src 16-byte aligned.
 * rfx_fm_cm_affine: to_fm = 1: out[n][f][bin][c] = in[n][c][bin][f] a[n] + b[n]; to_fm = 0: out[n][c][bin][f] = in[n][f][bin][c] a[n] + b[n]
 *   (b may be NULL): the de-standardisation fused with the layout change in front of the inverse STFT, and its backward. */
int rfx_cl_im2col_fm(const float* src, const float* coef_a, const float* coef_b, int32_t N, int32_t F, int32_t bins, void* dst, void* stream);
int rfx_fm_cm_affine(const float* in, float* out, const float* coef_a, const float* coef_b, int32_t N, int32_t bins, int32_t F, int32_t to_fm,
                     void* stream);
/* out = g * gelu'(z) over n bf16 values of dense channels-last tensors (n % 8 == 0) */
int rfx_cl_dgelu(const void* g, const void* z, void* out, int64_t n, void* stream);
/* GLU backward on dense channels-last tensors: g [npos][C], zab [npos][2C] = stored [a | b] -> out [npos][2C] */
int rfx_cl_dglu(const void* g, const void* zab, void* out, int64_t npos, int32_t C, void* stream);

/* Weight gradients on channels-last operands, deterministic (no atomics):
 *   D[m][(r, t, c)] = sum over (n, oa, b) of P[n][oa][b][m] * Q[n][oa*SA + da0 + r][b + db0 + t*db_step][c]     (Q = 0 outside)
 * = dW of Conv2d / ConvTranspose2d with P, Q = (output gradient, input) or (input, output gradient).  rfx_cl_wgrad cuts the
 * positions into S splits x D tiles (32 RW rows x the (tap, channel) columns of a CW-channel slice of Q); every workgroup leaves
 * its fp32 accumulators in its own slot of `ws` (rfx_cl_wgrad_ws_floats floats).  rfx_cl_wgrad_reduce sums the slots in a fixed
 * order and writes / accumulates through `map` (host-built: flat weight index per accumulator cell, wn + m = bias gradient m,
 * -1 = none) -- replaces the fp32 atomicAdd weight gradients of csrc/gemm_wgrad.h for the layers on this layout.
 * Call sites replaced: torch autograd of nn.Conv2d / nn.ConvTranspose2d weights inside torchaudio HDemucs (remfx/models.py:308,317). */
typedef struct rfx_cl_wgrad_desc {
  rfx_cl_tensor p, q;
  int32_t N, OA, IA, B;          /* rows of P, rows of Q, positions per row (B % 64 == 0) */
  int32_t SA, da0, NTR;          /* row taps: Q row oa*SA + da0 + r, r < NTR, SA <= NTR */
  int32_t NTC, db0, db_step;     /* column taps, |db| <= 8 */
  int32_t M, Cq, CW;             /* D rows (channels of P from p.c0); channels of Q (from q.c0); Q channels per D tile (% 16 == 0) */
  int32_t RW, WK;                /* 32-row tiles per D tile (2 | 3); K split inside a workgroup (1 | 2 | 4) */
  int32_t S, ahead, bias;        /* position splits (every split non-empty); prefetch distance in steps; 1 = also sum P over positions */
  int32_t PW;                    /* positions per step: 64 | 128 (B % PW == 0) */
  float* ws;
} rfx_cl_wgrad_desc;
int64_t rfx_cl_wgrad_ws_floats(const rfx_cl_wgrad_desc* d);
int rfx_cl_wgrad(const rfx_cl_wgrad_desc* d, void* stream);
int rfx_cl_wgrad_reduce(const float* ws, const int32_t* map, int64_t nmap, int32_t S, int32_t DT, int32_t RW, int32_t WK, float* dw,
                        int64_t wn, float* db, int32_t accumulate, void* stream);

/* Fused DConv depth-layer on channels-last samples (csrc/cl_dconv.hip): torchaudio HDemucs `_DConv` layers of the frequency
 * encoder (reference call site remfx/models.py:319), samples x [S][256][C] bf16 (S = clips x frequency rows), H = C / 4:
 *   h = conv1d(x; W1, b1, dilation dil); a = GELU(GN(1, H)(h)); z = conv1d(a; W2, b2); y = x + scale * GLU(GN(1, 2C)(z)).
 * forward: y; with a / hpre / stats non-NULL (training) also a and h as [S][256][Hp] bf16 (Hp = H rounded up to 16, pad channels 0)
 * and (mean1, rstd1, mean2, rstd2) per sample.  backward: dx into y, dz [S][256][2C] and dh [S][256][Hp] for the two
 * weight-gradient GEMMs (rfx_cl_wgrad), parameter-gradient partials per workgroup into `partial` ([grid][5C + 2H]) and their
 * fixed-order sum into pgrad (dscale[C] | dgn2w[2C] | dgn2b[2C] | dgn1w[H] | dgn1b[H]).  w*p: MFMA fragments packed by
 * rfx_cl_pack from the index tables of remfx_amd/cldconv.py.  rfx_cl_dconv_ok: shapes the kernels take. */
typedef struct rfx_cl_dconv_desc {
  const void* x;           /* forward input */
  const void* gy;          /* backward: gradient of y */
  void* y;                 /* forward: y; backward: dx */
  void* a; void* hpre; float* stats;
  void* dz; void* dh; float* partial;
  const void* w1p; const void* w2p; const void* w2dp; const void* w1dp;
  const float *b1, *g1w, *g1b, *b2, *g2w, *g2b, *scale;
  int32_t S, C, H, dil, grid, x_or_gy_ok;
  float eps;
  int32_t TPS;             /* 256-position tiles per sample (S counts TILES): 1 = the frequency branch; > 1 = the time branch, whose GroupNorm
                            * spans a whole clip -- forward in three passes with the statistics reduced in between (stats and `partial`,
                            * 2 floats per tile, required) */
  float* tsum; float* sums; /* backward in passes (TPS > 1 or C = 96): tile sums [S][2], per-sample means [S / TPS][4]; dz / dh come out final,
                            * dx is left to the caller (rfx_cl_conv: gy + the transposed 3-tap convolution of dh) */
  float* pg_dst[5];        /* backward, optional (all five or none): where dscale[C], dgn2w[2C], dgn2b[2C], dgn1w[H], dgn1b[H] are ADDED
                            * (the parameters' slices of a flat gradient buffer) instead of being written to pgrad */
} rfx_cl_dconv_desc;
int rfx_cl_dconv_ok(int32_t C, int32_t H, int32_t T, int32_t backward);
int rfx_cl_dconv_fwd(const rfx_cl_dconv_desc* d, void* stream);
int rfx_cl_dconv_bwd(const rfx_cl_dconv_desc* d, float* pgrad, void* stream);

/* out[a][c] (+)= scale * sum over (n, b) of x[n][a][b][c], deterministic (G fixed-order partials of A * C floats each in `partial`):
 * bias gradients (A = 1, rows folded into N) and the frequency-embedding gradient of Hybrid Demucs. */
int rfx_cl_rowsum(const rfx_cl_tensor* x, int32_t N, int32_t A, int32_t B, int32_t C, int32_t G, float scale, float* partial, float* out,
                  int32_t accumulate, void* stream);

int rfx_abi_version(void);
/* channel tiles per wave the MFMA forward kernel should use for M output rows and reduction length K
 * (0 = thin path; short-K, output-bound problems get R = 1 for occupancy);
 * the packed A matrix must have Mpad = ceil(M / (32*R)) * 32*R  (R=0: Mpad = 8). */
int rfx_gemm_pick_r(int32_t M, int32_t K);

#ifdef __cplusplus
}
#endif
#endif /* REMFX_HIP_H */
