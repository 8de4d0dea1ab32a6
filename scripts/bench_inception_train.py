"""Inception-v3 full fine-tune throughput (BASELINE config 5) on one GPU: native executor vs PyTorch eager.

    python scripts/bench_inception_train.py --batch 64 --steps 10 [--eager] [--optimizer sgd|adam]

Synthetic 299x299 images, random-init weights; the timed region is the complete step (forward with batch-norm
statistics, loss, backward, optimizer).  Prints one JSON line per engine.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--eager", action="store_true", help="also time the PyTorch eager (MIOpen, bf16 autocast-free fp32/bf16) step")
    ap.add_argument("--eager-dtype", default="bf16")
    ap.add_argument("--core", default="auto", choices=["auto", "v1", "v2"], help="conv GEMM core selection")
    args = ap.parse_args()
    from distributed_tensorflow_amd.models.inception_train import InceptionTrainer, torch_forward_train
    from distributed_tensorflow_amd.models.inception_v3 import graph_flops
    B = args.batch
    torch.manual_seed(0)
    x = torch.zeros(B, 299, 299, 8, device="cuda")
    x[..., :3] = torch.randn(B, 299, 299, 3, device="cuda")
    x = x.to(torch.bfloat16)
    labels = torch.randint(0, args.classes, (B,), device="cuda", dtype=torch.int32)
    tr = InceptionTrainer(args.classes, B, "cuda", lr=0.01, optimizer=args.optimizer, seed=1, conv_core=args.core)
    print(tr.tuning_summary(), flush=True)
    for _ in range(args.warmup):
        tr.train_step(x, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_step(x, labels)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    fl = 3.0 * graph_flops(tr.graph) * B
    print(json.dumps({"engine": "native", "batch": B, "ms_per_step": round(dt * 1e3, 2), "images_per_sec": round(B / dt, 1),
                      "model_tflops": round(fl / dt / 1e12, 1), "loss": tr.metrics()[0], "optimizer": args.optimizer, "conv_core": args.core,
                      "activation_gb": round(tr.activation_bytes() / 1e9, 2), "params": tr.num_params}), flush=True)
    if args.eager:
        dt_ = torch.bfloat16 if args.eager_dtype == "bf16" else torch.float32
        params = {k: tr.layout.view(tr.P, k).detach().clone().requires_grad_(True) for k in tr.layout.entries}
        opt = torch.optim.SGD(list(params.values()), lr=0.01) if args.optimizer == "sgd" else torch.optim.Adam(list(params.values()), lr=1e-4)
        xe = x[..., :3].permute(0, 3, 1, 2).to(dt_).contiguous(memory_format=torch.channels_last)
        del tr
        torch.cuda.empty_cache()

        def step():
            opt.zero_grad(set_to_none=True)
            cast = {k: (v.to(dt_) if k.endswith("conv2d_params") else v) for k, v in params.items()}
            feat = torch_forward_train(tr_graph, cast, xe, channels_last=True)
            logits = feat @ params["final_training_ops/weights/final_weights"] + params["final_training_ops/biases/final_biases"]
            loss = torch.nn.functional.cross_entropy(logits, labels.long())
            loss.backward()
            opt.step()
            return loss
        from distributed_tensorflow_amd.models.inception_v3 import build_graph
        tr_graph = build_graph()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"engine": "pytorch-eager-" + args.eager_dtype, "batch": B, "ms_per_step": round(dte * 1e3, 2),
                          "images_per_sec": round(B / dte, 1), "model_tflops": round(fl / dte / 1e12, 1), "loss": float(loss)}),
              flush=True)


if __name__ == "__main__":
    main()
