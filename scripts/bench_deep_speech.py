"""Times the DeepSpeech training step (shipped recipe sizes of examples/asr/hkust/configs/deep_speech.json:
64 filters, six GRU layers of 1680 units, 40-dim fbank) on synthetic data:
    python scripts/bench_deep_speech.py [--batch 32] [--frames 800] [--steps 5] [--layers 6] [--eager]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--labels", type=int, default=20)
    ap.add_argument("--vocab", type=int, default=3650)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=1680)
    ap.add_argument("--rnn", default="cudnngru")
    ap.add_argument("--eager", action="store_true")
    a = ap.parse_args()
    from athena_amd import BaseSolver, DeepSpeechModel, WarmUpAdam
    from athena_amd.data import DataDescriptions, make_samples
    cfg = {"conv_filters": 64, "rnn_hidden_size": a.hidden, "num_rnn_layers": a.layers, "rnn_type": a.rnn}
    model = DeepSpeechModel(DataDescriptions(a.vocab, 40), cfg, precision="bf16", seed=1)
    solver = BaseSolver(model, WarmUpAdam({"d_model": 512, "warmup_steps": 8000, "k": 0.5}),
                        config={"clip_norm": 100.0, "enable_tf_function": not a.eager, "log_interval": 10 ** 9})
    s = model.prepare_samples(make_samples(a.batch, a.frames, 40, a.vocab + 1, a.labels, seed=2))
    for _ in range(a.warmup):
        loss = solver.train_step(s)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = solver.train_step(s)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"workload": "deep_speech b%d t%d h%d x%d %s" % (a.batch, a.frames, a.hidden, a.layers, a.rnn),
                      "ms_per_step": dt * 1e3, "audio_sec_per_sec": a.batch * a.frames * 0.01 / dt,
                      "loss": float(loss), "hip_graph": not a.eager}))


if __name__ == "__main__":
    main()
