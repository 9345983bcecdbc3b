"""Benchmark of the ViTa Stage-I MAE train step (BASELINE.json configs[1]) on MI355X.

  python bench.py --gpus N --steps K --warmup W [--frames T] [--batch B] [--workload stage1|stage2|seg_encoder]

The default (stage1, T=50, B=2) is the configuration BASELINE.json's metric is quoted on.  The other workloads are
SURVEY 8(d)'s cfg4 (Stage II contrastive step: T=5, mask 0.5, 32 subjects/GPU, activation checkpointing as in the
yaml) and cfg5 (unmasked encoder with skip connections, N=23 041, fwd+bwd) through the same kernels.

A step = forward + backward + gradient all-reduce (N>1) + Adam over one batch of synthetic
SAX+LAX cine stacks already resident in HBM.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share one serialise.
# The step uses the main stream, the weight-gradient stream, the reducer's stream and RCCL's own: give them room.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

REF_CFG = dict(  # configs/imaging_model/pretraining_reconstruction_mae.yaml + configs/imaging_params.py defaults
    patch_size=(5, 8, 8), enc_embed_dim=1025, enc_depth=6, enc_num_heads=5, dec_embed_dim=1025, dec_depth=2,
    dec_num_heads=5, mlp_ratio=4.0, mask_ratio=0.7, mask_type="random", use_enc_pe=True, circular_pe=False,
    patch_embed_cls="PatchEmbed", patch_in_channels=1, mask_loss=False, grad_checkpointing=False)
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def blk_flops(n, d):
    """one Block forward per sample."""
    return 24.0 * n * d * d + 4.0 * n * n * d


def step_flops_per_subject(T, cfg=REF_CFG, S=9, HW=128):
    L = S * (T // cfg["patch_size"][0]) * (HW // cfg["patch_size"][1]) * (HW // cfg["patch_size"][2])
    keep = int(L * (1 - cfg["mask_ratio"]))
    D, Dd, P = cfg["enc_embed_dim"], cfg["dec_embed_dim"], 320
    fwd = 2.0 * keep * P * D + cfg["enc_depth"] * blk_flops(keep + 1, D) + 2.0 * (keep + 1) * D * Dd \
        + cfg["dec_depth"] * blk_flops(L + 1, Dd) + 2.0 * (L + 1) * Dd * P
    return 3.0 * fwd, L, keep


# n_classes of the 32 multi-class tabular features (selected_feaure_names.py:109-140)
TAB_MULTI = [3, 7, 6, 6, 6, 6, 6, 4, 3, 3, 5, 5, 6, 3, 3, 8, 8, 8, 3, 6, 7, 8, 7, 8, 8, 8, 7, 8, 3, 4, 3, 3]
TAB_NUM, TAB_SINGLE = 67, 18


def stage2_flops_per_subject(T, ckpt, D=1025, mask_ratio=0.5):
    """fwd = kept-patch embedding + 6 encoder blocks + 2 tabular blocks + the two heads; step = 3x fwd, 4x for the
    checkpointed blocks."""
    L = 9 * (T // 5) * 16 * 16
    keep = int(L * (1 - mask_ratio))
    nt = 1 + TAB_NUM + TAB_SINGLE + len(TAB_MULTI)
    blocks = 6 * blk_flops(keep + 1, D) + 2 * blk_flops(nt, D)
    rest = 2.0 * keep * 320 * D + 2 * (2.0 * D * D + 2.0 * D * 128)
    return (4.0 if ckpt else 3.0) * blocks + 3.0 * rest, L, keep


def seg_flops(T, ckpt, D=1025):
    L = 9 * (T // 5) * 16 * 16
    return (4.0 if ckpt else 3.0) * 6 * blk_flops(L + 1, D) + 3.0 * 2.0 * L * 320 * D, L


CFG1 = dict(D=129, heads=3, enc_depth=2, dec_depth=1, B=2, T=50)   # BASELINE.json configs[0]; see cpu_baseline


def cpu_threads_available():
    """Threads this process may really use: the scheduler affinity, capped by OMP_NUM_THREADS (the GPU box exports its
    CPU share there; os.cpu_count() shows the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, n)


def cpu_baseline(steps=3):
    """BASELINE.md section 2 'CPU baseline plan': the oracle (CPU restatement of the reference path: conv3d, rand/argsort/gather
    mask, materialised-softmax Blocks, MSE, Adam; fp32 stock torch CPU ops) timed on cfg1 ITSELF - Stage I, D=129, heads 3,
    encoder depth 2, B=2, T=50 (L=23 040, keep 6 912, N_dec=23 041), decoder 129 / 1 block / 3 heads (BASELINE.json names no
    decoder for the tiny config; one block keeps the three saved 23 041^2 fp32 score tensors per subject at ~42 GB of host
    memory) - 1 warm-up + `steps` timed steps at n = all available threads and again at n = 8.  No extrapolation: subjects/s
    = B / mean step time."""
    from oracle import vita_oracle as O
    import numpy as np
    c = CFG1
    img = (9, c["T"], 128, 128)
    p0 = O.init_stage1_params(img_shape=img, patch_size=REF_CFG["patch_size"], enc_dim=c["D"], enc_depth=c["enc_depth"],
                              enc_heads=c["heads"], dec_dim=c["D"], dec_depth=c["dec_depth"], dec_heads=c["heads"], seed=1)
    imgs = torch.rand(c["B"], *img, generator=torch.Generator().manual_seed(1))
    L = 9 * (c["T"] // 5) * 16 * 16
    cfg = dict(patch_size=REF_CFG["patch_size"], enc_depth=c["enc_depth"], enc_heads=c["heads"], dec_depth=c["dec_depth"],
               dec_heads=c["heads"], mask_ratio=0.7)
    out = {}
    n_all = cpu_threads_available()
    for n in dict.fromkeys([n_all, min(8, n_all)]):
        torch.set_num_threads(n)
        p, state, times = {k: v.clone() for k, v in p0.items()}, {}, []
        for i in range(steps + 1):
            noise = np.random.RandomState(i).rand(c["B"], L).astype(np.float32)
            t0 = time.perf_counter()
            O.stage1_train_step(imgs, p, noise, state, lr=1e-4, **cfg)
            times.append(time.perf_counter() - t0)
        out[n] = sum(times[1:]) / steps
    return out, n_all


def profiler_preloaded():
    """Non-empty (the evidence) when this process runs under a GPU profiler's preload (rocprofv3 / rocprof)."""
    pre = os.environ.get("LD_PRELOAD", "")
    if any(t in pre.lower() for t in ("rocprof", "roctracer", "rocprofiler")):
        return "LD_PRELOAD=" + pre
    for k in os.environ:
        if k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")):
            return k + " is set"
    return ""


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without an external launcher: this parent - which never touches the GPU - starts one
    rank per GPU through torch.distributed.run on 127.0.0.1 (the reference spawns its own ranks too: Lightning DDP,
    main.py:61), relays their output (rank 0 prints the JSON line) and exits with their status."""
    import socket
    import subprocess
    why = profiler_preloaded()
    if why:
        # A profiler's preloaded library initialises the GPU before main() runs (rocprofv3 --pmc does), and starting
        # the launcher from a GPU-initialised process is the exec hop this pool forbids.  Multi-rank profiling is the
        # other way round: the launcher outside, the profiler per rank -
        #   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --no-python \
        #       rocprofv3 --kernel-trace --stats -d out -- python3 bench.py --gpus N ...
        print(f"bench.py --gpus {n}: refusing to launch ranks from a profiled process ({why}); run the launcher outside "
              "and the profiler per rank (see bench.py launch_ranks)", file=sys.stderr, flush=True)
        return 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """VITA_BENCH_DRYRUN=1: the launcher / rendezvous / max-over-ranks plumbing with gloo and no GPU (CPU rehearsal)."""
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if os.environ.get("VITA_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        sys.exit(7)                                         # rehearses a rank that dies: the parent must exit non-zero
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "max_over_ranks": t.item(), "steps": args.steps}), flush=True)


# rocprofv3 kernel names behind each timed launch group of vita_amd.ops (for roofline.traffic)
GROUP_KERNELS = {"attn_bwd_kernels": ("attn_delta_kernel", "attn_bwd_dkdv_kernel", "attn_bwd_dkdv8_kernel", "attn_bwd_dkdv_p_kernel",
                                      "attn_bwd_dq_kernel", "attn_bwd_dq_ds_kernel", "attn_bwd_tail_reduce_kernel",
                                      "attn_bwd_dq_tail_reduce_kernel"),
                 "attn_fwd_kernel": ("attn_fwd_kernel", "attn_fwd16_kernel", "attn_fwd_tail_combine_kernel")}


def pmc_traffic(group, frames, batch, launches_per_step):
    """HBM bytes per launch of a timed group from the newest committed PMC summary (profiles/*_traffic.json, written by
    tools/profile_summary.py from separate FETCH_SIZE / WRITE_SIZE passes of this same command; FETCH_SIZE doubled as
    the gfx950 guide prescribes).  None when no summary matches this workload."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        t = json.load(open(f))
        if t.get("frames") != frames or t.get("per_gpu_batch") != batch:
            continue
        names = GROUP_KERNELS.get(group)
        if names is None:       # GEMM groups mix two kernels and several tile variants per layout: not attributed
            return None, None
        tot, steps = 0.0, None
        for k, v in t["kernels"].items():
            if any(k.startswith(n) or (" " + n) in k for n in names):
                tot += ((v["read_bytes_per_launch"] or 0) + (v["write_bytes_per_launch"] or 0)) * v["launches"]
        steps = t.get("steps_profiled")
        if tot and steps and launches_per_step:
            return tot / steps / launches_per_step, os.path.basename(f)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="stage1", choices=["stage1", "stage2", "seg_encoder"])
    ap.add_argument("--frames", type=int, default=None, help="T: stage1 50 = BASELINE.json shape, 25 = shipped yaml; stage2 5")
    ap.add_argument("--batch", type=int, default=None, help="subjects per GPU (yaml: stage1 2, stage2 256 global = 32 x 8)")
    ap.add_argument("--no-ckpt", action="store_true", help="stage2 / seg_encoder without activation checkpointing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps per thread setting (after 1 warm-up)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--materialized-attention", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))     # before anything touches the GPU in this process
    if os.environ.get("VITA_BENCH_DRYRUN") == "1":
        return dry_run(args)

    if os.environ.get("VITA_BENCH_WATCHDOG"):               # seconds: dump every thread's stack and exit if still running
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VITA_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    # VITA_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, bucketed all-reduce, barriers) with one rank too
    use_dist = world > 1 or os.environ.get("VITA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's kernels on a high-priority stream: a hardware queue of their own (never the main stream's) and first
        # pick of the CUs, so a bucket's all-reduce proceeds under the backward instead of behind it
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from vita_amd import nn as vnn, ops
    from vita_amd.train import BucketedAllReduce, FlatAdam, ReconMAE, stage1_train_step
    if args.materialized_attention:
        vnn.DEFAULT_FLASH = False

    wl = args.workload
    T = args.frames if args.frames is not None else {"stage1": 50, "stage2": 5, "seg_encoder": 50}[wl]
    B = args.batch if args.batch is not None else {"stage1": 2, "stage2": 32, "seg_encoder": 1}[wl]
    ckpt = not args.no_ckpt
    img_shape = [9, T, 128, 128]
    torch.manual_seed(1)                      # reference seed (yaml general.seed / main.py:22), same on every rank
    gen = torch.Generator(device="cuda").manual_seed(1 + rank)
    imgs = torch.rand(B, *img_shape, device="cuda", generator=gen)
    if wl == "stage1":
        model = ReconMAE(img_shape=img_shape, use_both_axes=True, **REF_CFG).cuda()
        opt = FlatAdam(model.parameters(), lr=1e-4)
        reducer = BucketedAllReduce(opt) if use_dist else None
        step = lambda: stage1_train_step(model, opt, imgs, reducer)
    elif wl == "stage2":
        from types import SimpleNamespace
        from vita_amd.train import MultimodalSimCLR, stage2_train_step, warmup_cosine_lr
        imaging = dict(REF_CFG, mask_ratio=0.5, mask_loss=True, grad_checkpointing=ckpt, projection_dim=128)
        tab = SimpleNamespace(grad_checkpointing=ckpt, embedding_dim=1025, encoder_num_heads=5, encoder_mlp_ratio=4.0,
                              encoder_num_layers=2, projection_dim=128)
        feats = {"numerical": [f"n{i}" for i in range(TAB_NUM)], "single_categorical": [f"s{i}" for i in range(TAB_SINGLE)],
                 "multi_categorical": {f"m{i}": [n] for i, n in enumerate(TAB_MULTI)}}
        model = MultimodalSimCLR(img_shape, imaging, tab, feats, temperature=0.1, lambda_0=0.5,
                                 gather_distributed=use_dist).cuda()
        # a forced one-rank distributed run exercises the all-gather / reduce-scatter / loss all-reduce too
        model.criterion.collectives_at_world1 = use_dist and world == 1
        opt = FlatAdam(model.parameters(), lr=3e-3, weight_decay=1e-4)
        reducer = BucketedAllReduce(opt) if use_dist else None
        tabs = torch.cat([torch.randn(B, TAB_NUM, device="cuda", generator=gen),
                          torch.randint(0, 2, (B, TAB_SINGLE), device="cuda", generator=gen).float(),
                          torch.stack([torch.randint(0, n, (B,), device="cuda", generator=gen) for n in TAB_MULTI], 1).float()], 1)
        lr = warmup_cosine_lr(3e-3, 10, 10, 200)          # the schedule's peak (epoch = warmup_epochs)
        step = lambda: stage2_train_step(model, opt, imgs, tabs, reducer, lr=lr)[0]
    else:
        from vita_amd.networks import ImagingMaskedEncoder
        model = ImagingMaskedEncoder(img_shape=img_shape, use_both_axes=True,
                                     **dict(REF_CFG, mask_ratio=0.0, grad_checkpointing=ckpt)).cuda()
        opt = FlatAdam(model.parameters(), lr=5e-4)
        reducer = BucketedAllReduce(opt) if use_dist else None

        def step():                                        # SURVEY cfg5: sum of the output and hidden states 2, 4 as the loss
            opt.zero_grad()
            y, hidden = model.forward_with_skip_connection(imgs)
            loss = y.float().mean() + hidden[2].float().mean() + hidden[4].float().mean()
            loss.backward()
            if reducer is not None:
                reducer.finish()
            opt.step(grad_scale=1.0 / world)
            return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_kernel_timing:
        ops.PROF = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_issued = time.perf_counter() - t0       # host done issuing; close to dt when the host, not the GPU, sets the pace
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROF = ops.PROF, None
    if use_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        if wl == "stage1":
            step_fl, L, keep = step_flops_per_subject(T)
            metric = "CMR subjects/sec per train step (Stage I MAE)"
            desc = (f"Stage I MAE train step (fwd+bwd+Adam), D=1025 enc6/dec2 heads5 mask0.7, "
                    f"9x{T}x128x128 SAX+LAX, patch 5x8x8 (L={L}, keep={keep})")
        elif wl == "stage2":
            step_fl, L, keep = stage2_flops_per_subject(T, ckpt)
            metric = "CMR subjects/sec per train step (Stage II CLIP)"
            desc = (f"Stage II contrastive train step (imaging D=1025 enc6 mask0.5 + TabularViT 117 features depth2 + SimCLR heads "
                    f"+ CLIP loss + Adam wd1e-4), 9x{T}x128x128 (L={L}, keep={keep}), activation checkpointing {'on' if ckpt else 'off'}"
                    + (", global negatives over all ranks" if world > 1 else ""))
        else:
            step_fl, L = seg_flops(T, ckpt)
            keep = L
            metric = "CMR subjects/sec, unmasked encoder fwd+bwd with skip connections (cfg5)"
            desc = (f"ImagingMaskedEncoder.forward_with_skip_connection fwd+bwd+Adam, D=1025 depth6 heads5, N={L + 1}, "
                    f"activation checkpointing {'on' if ckpt else 'off'}")
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": metric, "value": round(value, 4), "unit": "subjects/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": desc,
                       "per_gpu_batch": B, "global_batch": world * B, "frames": T,
                       "parallelism": f"dp{world}", "loss": round(float(loss.detach()), 6)},
            "step_tflop_per_subject": round(step_fl / 1e12, 3),
            "model_tflops_per_gpu": round(step_fl * B * args.steps / dt / 1e12, 1),
            "host_issue_ms_per_step": round(t_issued / args.steps * 1e3, 3),
        }
        if prof:
            agg = {}
            for name, e0, e1, work in prof:
                a = agg.setdefault(name, [0.0, 0.0, 0])
                a[0] += e0.elapsed_time(e1) * 1e-3; a[1] += work; a[2] += 1
            dom = max(agg, key=lambda k: agg[k][0])
            tsec, work, cnt = agg[dom]
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(work / tsec / 1e12, 1),
                               "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(work / tsec / 1e12 / PEAK_BF16_TFLOPS, 4),
                               "traffic": None, "launches_per_step": cnt // args.steps,
                               "avg_launch_ms": round(tsec / cnt * 1e3, 4),
                               "algorithmic_gflop_per_launch": round(work / cnt / 1e9, 2)}
            tr, src = pmc_traffic(dom, T, B, cnt // args.steps) if wl == "stage1" else (None, None)
            if tr is not None:
                out["roofline"]["traffic"] = round(tr)
                out["roofline"]["traffic_unit"] = "HBM bytes per launch"
                out["roofline"]["traffic_source"] = "profiles/" + src
            out["kernels"] = {k: {"ms_per_step": round(v[0] / args.steps * 1e3, 3),
                                  "tflops": round(v[1] / v[0] / 1e12, 1),
                                  "mfma_frac": round(v[1] / v[0] / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "launches_per_step": v[2] // args.steps}
                              for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
            out["kernels_note"] = ("HIP-event time per launch group; weight-gradient GEMMs (<1,1>) run on a second stream beside the "
                                   "dgrad GEMMs (<0,1>) in the backward, so those two groups overlap and their times are not additive")
        if world == 1 and not args.no_cpu_baseline and wl == "stage1":
            times, n_all = cpu_baseline(args.cpu_steps)
            c = CFG1
            out["cpu_baseline"] = {
                "value": round(c["B"] / times[n_all], 5), "unit": "subjects/s", "cores": n_all, "kind": "port",
                "by_threads": {str(n): {"s_per_step": round(t, 3), "subjects_per_s": round(c["B"] / t, 5)} for n, t in times.items()},
                "sample": (f"BASELINE.json configs[0] itself, measured (no extrapolation): oracle Stage-I train step (fwd+bwd+Adam), "
                           f"D={c['D']} heads {c['heads']} enc depth {c['enc_depth']}, decoder {c['D']}/{c['dec_depth']} block/{c['heads']} heads "
                           f"(BASELINE names no decoder; 1 block bounds host memory at ~42 GB), B={c['B']}, T={c['T']} (L=23040, keep 6912, "
                           f"N_dec=23041), fp32 torch CPU ops with materialised 23041^2 attention; 1 warm-up + {args.cpu_steps} timed steps "
                           f"per thread setting; torch.set_num_threads = {' and '.join(str(n) for n in times)}; a different (tiny) model than "
                           f"the GPU headline config, as BASELINE.md section 2 prescribes")}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
