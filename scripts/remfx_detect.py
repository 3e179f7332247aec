"""Single-file detect-and-remove, same command line as the reference scripts/remfx_detect.py:13-61 (`remfx_detect.sh`):
    python scripts/remfx_detect.py +exp=remfx_detect +audio_input=in.wav +output_path=out.wav
No Trainer: the five effect-specific removal models and the Cnn14 detector are instantiated from cfg.ckpts /
cfg.classifier, their checkpoints loaded strictly, the file is decoded, resampled to cfg.sample_rate ON THE DEVICE
(remfx_amd.resample: the polyphase filter bank of torchaudio.transforms.Resample as one strided gather-GEMM), mixed to
mono, run through RemFXChainInference.forward(batch, 0, verbose=True) at its full length (any length: the attention kernels stream
the keys beyond 256 frames), and written as a float32 WAV (torchaudio.save's default for float tensors)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from remfx_amd import config as rcfg  # noqa: E402
from remfx_amd.datasets import load_wav, save_wav  # noqa: E402
from remfx_amd.resample import resample  # noqa: E402
from scripts.chain_inference import build  # noqa: E402


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = rcfg.compose(os.environ.get("REMFX_CFG_DIR", os.path.join(ROOT, "cfg")), "config.yaml", argv)
    if not torch.cuda.is_available():
        raise RuntimeError("remfx_detect needs the GPU: the removal networks have no CPU path")
    device = torch.device("cuda", 0)
    print("Loading models...")
    inference_model = build(cfg, device)                       # remfx_detect.py:15-41
    audio_file = cfg["audio_input"]
    print("Loading", audio_file)
    audio, sr = load_wav(audio_file)                           # (channels, samples) float32, remfx_detect.py:45
    audio = audio.to(device)
    if sr != cfg["sample_rate"]:
        audio = resample(audio, sr, cfg["sample_rate"])                             # :47
    audio = audio.mean(0, keepdim=True).unsqueeze(0)           # mono + batch dim, :49-51
    batch = [audio, audio, None, None]
    _, y = inference_model(batch, 0, verbose=True)             # :55
    output_path = cfg["output_path"] if "output_path" in cfg else "./output.wav"
    print("Saving output to", output_path)
    save_wav(output_path, y[0].cpu(), cfg["sample_rate"])
    return output_path


if __name__ == "__main__":
    main()
