#!/usr/bin/env python3
"""Headline benchmark: slates/s, forward+backward(+optimizer), approxNDCG + DoubleLayerNet, slate 128 x 136
features, fp32, synthetic MSLR-WEB30K-shaped data resident in HBM (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = one optimizer step of the query-sharded trainer on `--batch` slates per GPU: ONE fused launch
(scorer fwd -> approxNDCG -> scorer bwd -> weight gradients), a fixed-order gradient reduce, one all-reduce
of the flat [grads | loss] buffer (N > 1) and a fused Adam update.  Per-GPU work is fixed as N grows (weak).
Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused slate pipeline), timed
with HIP events on its own stream inside the timed region; `cpu_baseline` is the oracle's op-level
restatement of the reference's CPU path, timed on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd"))

F = 136
PEAK_HBM_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured float4 copy
PEAK_HBM_MEASURED_GBPS = 6290.0
PEAK_F32_MFMA_TFLOPS = 157.3    # dense fp32 MFMA (= fp32 vector) peak


def flops_per_doc(net):
    """Algorithmic FLOPs fwd + bwd (no dX), SURVEY.md section 8(a)/(d)."""
    if net == "double":
        mac = (136 * 136 * 2 + 136) + (136 * 136 * 3 + 136 * 2)      # 92 888
    elif net == "two64":
        mac = (136 * 64 + 64) + (136 * 64 + 64)                      # 17 536: fwd + (dW1, dw3); no dX, no dh1
    else:
        mac = (136 * 64 + 64 * 32 + 32) + (136 * 64 + 64 * 32 * 2 + 32 * 2)
    return 2.0 * mac


def pmc_traffic_per_launch(net, batch):
    """HBM bytes per launch of the pipeline kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate passes; on gfx950 FETCH_SIZE counts wide coalesced reads at half
    their size and is doubled, MI355X_MICROARCH.md section HBM).  None if no matching profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        lib_tag = "_f16x2" if "f16x2" in os.environ.get("LTR_LIB", "") else ""
        e = rec.get(f"{net}_b{batch}{lib_tag}")
        if e:
            return int((2.0 * e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024)
    except (OSError, ValueError, KeyError):
        pass
    return None


def synth(Q, S, device, seed):
    """X ~ N(0,1) (MSLR 'Norm' features are per-query normalised), grades with MSLR-like skew."""
    gen = torch.Generator(device=device).manual_seed(seed)
    X = torch.empty((Q, S, F), dtype=torch.float32, device=device)
    step = max(1, 8192 // max(S // 32, 1))
    for i in range(0, Q, step):
        X[i:i + step].normal_(generator=gen)
    p = torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01], device=device)
    y = torch.multinomial(p, Q * S, replacement=True, generator=gen).view(Q, S).float()
    return X, y


def cpu_baseline(net_kind, S, budget_s=12.0):
    """The reference's CPU path as restated by the oracle (torch CPU ops + autograd, [B,S,S] intermediates),
    B = 200 slates per step (main.py:59), timed on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ltr_oracle as O   # the timed CPU baseline ("port"), never part of the product path
    torch.manual_seed(2020)
    # B = 200 x 128 x 128 pair tensors oversubscribe a 128-thread box (round 2 measured it SLOWER there than on the build
    # container's 8 threads): cap the intra-op pool at 16 threads and say so
    threads_before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, threads_before)))
    B = 200
    if net_kind == "double":
        shapes = {"fc1.weight": (136, 136), "fc1.bias": (136,), "fc2.weight": (136, 136), "fc2.bias": (136,),
                  "fc3.weight": (1, 136), "fc3.bias": (1,)}
    elif net_kind == "two64":
        shapes = {"fc1.weight": (64, 136), "fc1.bias": (64,), "fc4.weight": (1, 64), "fc4.bias": (1,)}
    else:
        shapes = {"l1.weight": (64, 136), "l1.bias": (64,), "l2.weight": (32, 64), "l2.bias": (32,),
                  "l3.weight": (1, 32), "l3.bias": (1,)}
    p = {k: (torch.randn(v) * 0.05).requires_grad_(True) for k, v in shapes.items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-3)
    x = torch.randn(B, S, F)
    y = torch.randint(0, 5, (B, S)).float()

    def one():
        opt.zero_grad()
        if net_kind == "double":
            k1 = (torch.rand(B, S, 136) < 0.5).float()
            k2 = (torch.rand(B, S, 136) < 0.5).float()
            s = O.double_layer_forward(x, p, k1, k2)
        elif net_kind == "two64":
            s = O.two_layer_forward(x, p)
        else:
            s = O.triple_layer_forward(x, p)
        loss = O.approx_ndcg(s.squeeze(-1), y)
        loss.backward()
        opt.step()

    one()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        one()
        n += 1
    dt = time.perf_counter() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(threads_before)
    return {"value": round(B * n / dt, 1), "unit": "slates/s", "cores": used, "kind": "port",
            "sample": f"{n} steps of B=200 x S={S} x F=136 ({net_kind} net + approxNDCG + Adam, fp32) in {dt:.1f}s on {used} of "
                      f"{os.cpu_count()} host threads (torch intra-op pool capped at 16: the batch oversubscribes more); "
                      "oracle restatement of the reference's torch-CPU path (no dX, unlike the reference)"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this parent has made NO GPU call (it only parsed arguments);
    it starts the N ranks as a child `torch.distributed.run` (never exec from a GPU-touched process), relays
    their output -- rank 0's JSON line included -- and exits with the children's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.run(cmd, env=env).returncode
    sys.exit(rc)


def secondary_lines(a):
    """Second lines measured by CHILD processes with the same shapes (never exec from this GPU-touched process):
      f16x2  : the f16 x 2 split-precision variant of the pipeline (same C ABI, LTR_LIB; parity-green at the fp32 bars) on the
               headline workload -- reported next to the exact-fp32 headline, not instead of it;
      triple : TripleLayerNet (136-64-32-1, tripleLayer.py:5-17) on the headline workload: its fused step runs folded (HBM-bound);
      two64  : the 136-64-1 two-layer scorer BASELINE.json configs[0] names, the configuration the 60 % HBM target was
               written for (exact-fp32 library); two64_f16x2: the same on the split-precision variant;
      config5: BASELINE.json configs[4] -- architeture/transformer.py scorer (make_model: FC 136->128, 6 encoder blocks,
               8 heads, d_ff 2048, dropout 0.1) + approxNDCG, slate 256, bf16 operands (tools/bench_encoder.py)."""
    import subprocess
    here = os.path.abspath(__file__)
    variant = os.path.join(ROOT, "nn-with-pytorch-personalized-losses_amd", "ltr_mi355x", "libltr_mi355x_f16x2.so")
    base = [sys.executable, here, "--steps", str(a.steps), "--warmup", str(a.warmup), "--queries", str(a.queries), "--slate",
            str(a.slate), "--batch", str(a.batch), "--no-cpu-baseline", "--no-extras"]
    runs = {"f16x2": (base + ["--net", "double"], {"LTR_LIB": variant} if os.path.exists(variant) else None),
            "triple": (base + ["--net", "triple"], {}),
            "two64": (base + ["--net", "two64"], {}),
            "two64_f16x2": (base + ["--net", "two64"], {"LTR_LIB": variant} if os.path.exists(variant) else None)}
    out = {}
    for name, (cmd, env_add) in runs.items():
        if env_add is None:
            out[name] = {"error": "variant library not built"}
            continue
        try:
            r = subprocess.run(cmd, env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
            out[name] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "workload": j["config"]["workload"],
                         "roofline": {k: j["roofline"][k] for k in ("bound", "achieved", "frac", "unit", "kernel_ms", "hbm_achieved_GBps", "hbm_frac_of_8TBps",
                                                                    "mfma_frac_of_layerwise_flops", "mfma_frac_of_executed_flops") if k in j["roofline"]}}
            if j["roofline"].get("traffic") is not None:
                out[name]["roofline"]["traffic"] = j["roofline"]["traffic"]
            if name in ("f16x2", "two64_f16x2"):
                out[name]["library"] = "libltr_mi355x_f16x2.so (LTR_LIB): fp32 operands as two f16 pieces on the f16 matrix cores, fp32 accumulation"
                out[name]["roofline"]["note"] = ("achieved / frac: algorithmic fp32 FLOP/s over the fp32-MFMA peak (157.3 TF) the exact library is "
                                                 "bound by -- this variant issues 4 f16 piece products per fp32 product on the 2.5 PF f16 pipe")
        except Exception as e:          # a secondary line must never take the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
    try:
        enc = os.path.join(ROOT, "tools", "bench_encoder.py")
        j = None
        for extra in (["--graph"], []):          # the whole step as one hipGraph (device-side dropout epoch); eager launches if capture fails
            r = subprocess.run([sys.executable, enc, "--batch", "256", "--steps", "10", "--warmup", "3"] + extra, capture_output=True,
                               text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if lines:
                j = json.loads(lines[-1])
                break
        if j is None:
            raise RuntimeError(r.stderr[-200:])
        out["config5"] = {"value": j["slates_per_s"], "unit": "slates/s", "ms_per_step": j["ms_per_step"], "workload": j["workload"],
                          "dtype": "bf16", "roofline": {"bound": "mfma", "achieved": j["tflops"], "peak": 2500.0, "unit": "TFLOP/s",
                                                        "frac": j["frac_of_bf16_mfma_peak"]}}
    except Exception as e:
        out["config5"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--queries", type=int, default=100_000, help="resident queries per GPU")
    ap.add_argument("--slate", type=int, default=128)
    ap.add_argument("--batch", type=int, default=25_000, help="slates per GPU per step")
    ap.add_argument("--net", choices=["double", "triple", "two64"], default="double",
                    help="two64 = the 136-64-1 two-layer variant BASELINE.json configs[0] names (bench-only)")
    ap.add_argument("--loss", choices=["approxNDCG", "listnet", "lambdaLoss"], default="approxNDCG",
                    help="headline metric is approxNDCG; lambdaLoss uses ndcgLoss2PP_scheme (main_batch_execution.py:135)")
    ap.add_argument("--eval-mode", action="store_true", help="no dropout (default: training mode, like the reference loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements attached to the default headline line (split-precision variant, 136-64-1 net)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a.gpus)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    import torch.distributed as dist
    # one process per GPU.  (Rehearsal on a box with fewer GPUs than ranks: ranks share cards and
    # LTR_DIST_BACKEND=gloo stands in for RCCL, which refuses two ranks on one device.)
    backend = os.environ.get("LTR_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from architeture.doubleLayer import DoubleLayerNet
    from architeture.tripleLayer import TripleLayerNet
    from ltr_mi355x.dp import QueryShardedTrainer, sync_parameters
    from ltr_mi355x.scorer import FusedRanker

    torch.manual_seed(2020)
    if a.net == "two64":
        from ltr_mi355x.extra_nets import TwoLayerNet
        net = TwoLayerNet(F).to(dev)
    else:
        net = (DoubleLayerNet(F) if a.net == "double" else TripleLayerNet(F)).to(dev)
    net.train(not a.eval_mode)
    sync_parameters(net)
    Q, S, B = a.queries, a.slate, min(a.batch, a.queries)
    X, y = synth(Q, S, dev, 2020 + rank)
    ranker = FusedRanker(net, loss=a.loss, **({"weighing_scheme": "ndcgLoss2PP_scheme"} if a.loss == "lambdaLoss" else {}))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    trainer = QueryShardedTrainer(ranker, opt)
    n_win = max(1, Q // B)

    def step(i):
        lo = (i % n_win) * B
        return trainer.step(X[lo:lo + B], y[lo:lo + B], global_batch=B * world)   # equal shards: no size collective

    for i in range(a.warmup):
        step(i)
    # --- timed region: barrier + synchronize on both sides, max over ranks
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    cevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)] if world > 1 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ranker.kernel_events = evs[i]
        if cevs:
            trainer.comm_events = cevs[i]
        loss = step(a.warmup + i)
    ranker.kernel_events = None
    trainer.comm_events = None
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0          # this rank's own clock over the K steps (before the closing barrier)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    kern_ms = sum(s.elapsed_time(e) for s, e in evs) / a.steps
    # per rank: its own step time, its pipeline-kernel time and the time of the ONE all-reduce per step (HIP events on the stream the
    # collective is enqueued on) -- so that a 1/2/4/8 curve can be read: where a rank's step exceeds kernel + Adam, the collective
    # (or waiting for the slowest rank inside it) is the difference
    comm_ms = sum(s.elapsed_time(e) for s, e in cevs) / a.steps if cevs else 0.0
    mine = torch.tensor([dt_local / a.steps * 1e3, kern_ms, comm_ms], dtype=torch.float64, device=dev)
    per_rank = [mine]
    if world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    print(json.dumps({"rank": rank, "ms_per_step_local": round(float(mine[0]), 4), "kernel_ms": round(float(mine[1]), 4),
                      "allreduce_ms": round(float(mine[2]), 4)}), file=sys.stderr, flush=True)
    one_launch = S in (32, 64, 128)
    if not one_launch:
        # three-launch path: the bracketed backward launch is only part of the work (it recomputes the forward),
        # so the roofline figures are taken over the whole step instead of over one kernel
        kern_ms = dt / a.steps * 1e3
    final_loss = float(loss)

    if rank == 0:
        slates_per_s = world * B * a.steps / dt
        fl_slate = flops_per_doc(a.net) * S
        by_slate = S * (F + 1) * 4
        ach_tf = fl_slate * B / (kern_ms * 1e-3) / 1e12
        ach_gb = by_slate * B / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": f"slates/sec fwd+bwd {a.loss} slate={S} feat={F}",
            "value": round(slates_per_s, 1), "unit": "slates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.loss} + {a.net}LayerNet ({dict(double='136-136-136-1', triple='136-64-32-1', two64='136-64-1')[a.net]}) "
                                   f"{'train-mode dropout' if net.training and a.net == 'double' else 'no dropout'}, "
                                   f"{Q} queries x slate {S} x {F} feat fp32 per GPU resident in HBM, "
                                   f"{B} slates per GPU per step, "
                                   f"{'one fused launch (fwd+loss+bwd)' if S in (32, 64, 128) else 'forward launch + loss kernel + backward launch'}"
                                   " + grad all-reduce + Adam",
                       "queries_per_gpu": Q, "slate": S, "features": F, "batch_per_gpu": B, "net": a.net,
                       "resident_GB_per_gpu": round(Q * S * (F + 1) * 4 / 1e9, 2), "step_windows": n_win,
                       "step_windows_visited": min(n_win, a.warmup + a.steps),
                       "last_window_starts_at_float": int(((min(n_win, a.warmup + a.steps) - 1) % n_win) * B) * S * F,
                       "parallelism": f"query-sharded dp{world}", "final_loss": round(final_loss, 6)},
            "roofline": {"bound": "mfma", "achieved": round(ach_tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": pmc_traffic_per_launch(a.net, B) if S == 128 else None,
                         "kernel": ("fcw_fused_kernel (feature-partitioned, csrc/ltr_fcw.h)" if a.net == "two64" else "slate_pipeline_kernel<MODE_FUSED>") if one_launch
                         else "whole step: pipeline<MODE_FWD> + loss kernel + pipeline<MODE_BWD>",
                         "kernel_ms": round(kern_ms, 4),
                         "flops_per_slate": fl_slate, "bytes_per_slate": by_slate,
                         "hbm_achieved_GBps": round(ach_gb, 1), "hbm_frac_of_8TBps": round(ach_gb / PEAK_HBM_GBPS, 4),
                         "hbm_frac_of_measured_copy": round(ach_gb / PEAK_HBM_MEASURED_GBPS, 4)},
        }
        if a.net == "triple" and one_launch and getattr(ranker, "fold", None) is not None:
            # TripleLayerNet runs FOLDED (l2 . l1 as one 136 -> 32 layer, tripleLayer.py:14-16 has no activation between them): the
            # kernel executes 2.46 x fewer multiply-adds than the layer-by-layer formulation, so the fp32-MFMA fraction of the
            # REFERENCE's flops can exceed 1 -- the bound that is left is HBM (X read once)
            exec_fl = 2.0 * ((136 * 32 + 32) + (136 * 32 + 32)) * S
            r = out["roofline"]
            r.update({"bound": "hbm", "achieved": r["hbm_achieved_GBps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": r["hbm_frac_of_8TBps"],
                      "kernel": "fcw_fused_kernel<TripleFolded> (csrc/ltr_fcw.h, two document-split copies of the 32 folded units)",
                      "mfma_frac_of_layerwise_flops": round(ach_tf / PEAK_F32_MFMA_TFLOPS, 4), "executed_flops_per_slate": exec_fl,
                      "mfma_frac_of_executed_flops": round(exec_fl * B / (kern_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)})
            out["config"]["workload"] = out["config"]["workload"].replace("(136-64-32-1)", "(136-64-32-1, l2.l1 folded to 136-32-1 inside the step)")
        out["per_rank"] = [{"rank": r, "ms_per_step_local": round(float(t[0]), 4), "kernel_ms": round(float(t[1]), 4),
                            "allreduce_ms": round(float(t[2]), 4)} for r, t in enumerate(per_rank)]
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.net, S)
        if world == 1 and not a.no_extras and not a.no_cpu_baseline and a.net == "double" and a.loss == "approxNDCG" \
                and not os.environ.get("LTR_LIB"):
            out["secondary"] = secondary_lines(a)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
