#!/usr/bin/env python
"""Headline benchmark: prompt-tuning images/sec (fwd+bwd), ViT-B/16, per-GPU batch 256, synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: MVLPT with the CoOp head (16 learnable text-context tokens, class token in
the `middle`, 100 classes, text length 77), ViT-B/16, batch 256 per GPU.  One step = `MVLPT.forward_backward`:
image tower forward (no backward is needed: CoOp has no visual prompts), text tower forward +
backward over all 100 class prompts, cosine logits, cross-entropy, prompt-gradient all-reduce, SGD update.
Weak scaling: every rank processes its own 256 images; no data-path collective except the 32 KB gradient
all-reduce.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16 MFMA peak of one MI355X (MI355X_MICROARCH.md)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "gemm_hbm_traffic.json")   # written from rocprofv3 --pmc passes


def algorithmic_gflop_per_image(arch, B_global, C, L_text, n_ctx, n_vpt, causal_half=True):
    """SURVEY.md §8(d) conventions: 1 MAC = 2 FLOP; frozen weights => no dW; image backward only with visual
    prompts; text tower only with text context, amortised over the global batch, causal attention at half rate."""
    g2 = arch.grid ** 2
    Lv = 1 + n_vpt + g2
    dv, dt = arch.vision_width, arch.transformer_width

    def tower(tokens, L, d, layers, bwd, causal):
        att_f = (2 if causal else 4) * L * d
        att_b = (4 if causal else 8) * L * d
        f = tokens * layers * (24 * d * d + att_f)
        b = tokens * layers * (24 * d * d + att_b) if bwd else 0
        return f + b

    img = tower(Lv, Lv, dv, arch.vision_layers, n_vpt > 0, False)
    img += 2 * g2 * 3 * arch.vision_patch_size ** 2 * dv + 2 * dv * arch.embed_dim
    txt = 0
    if n_ctx > 0:
        txt = C * (tower(L_text, L_text, dt, arch.transformer_layers, True, True) + 2 * dt * arch.embed_dim)
    head = 6 * arch.embed_dim * C
    return (img + head + txt / B_global) / 1e9


def cpu_baseline_images_per_sec(arch, sd, B, C, L, n_ctx, sample_images=8):
    """Oracle (CPU restatement, kind "port") on the host cores: image tower forward on a bounded sample of the
    batch (extrapolated linearly, it is per-image independent) + the FULL text tower fwd+bwd + head."""
    from oracle import clip_oracle as O
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    img = torch.randn(sample_images, 3, arch.image_resolution, arch.image_resolution)
    name_lens = [1 + (i % 3) for i in range(C)]
    layout = O.build_prompt_layout(name_lens, n_ctx, L, "middle")
    eot = torch.tensor([n_ctx + nl + 2 for nl in name_lens])
    prefix = torch.randn(C, 1, arch.transformer_width) * 0.02
    suffix = torch.randn(C, L - 1 - n_ctx, arch.transformer_width) * 0.02
    ctx = torch.randn(n_ctx, arch.transformer_width) * 0.02
    with torch.no_grad():
        t0 = time.perf_counter()
        feat, _ = O.image_encoder_fwd(sd, img, None, None, heads=arch.vision_heads, need_bwd=False)
        t_img = time.perf_counter() - t0
        t0 = time.perf_counter()
        prompts = O.assemble_prompts(ctx, prefix, suffix, layout)
        txt, tctx = O.text_encoder_fwd(sd, prompts, eot, heads=arch.transformer_heads, need_bwd=True)
        full = feat.repeat((B + sample_images - 1) // sample_images, 1)[:B]
        logits, lctx = O.logits_fwd(full, txt, float(sd["logit_scale"].exp()))
        loss, dl = O.cross_entropy_fwd_bwd(logits, torch.randint(0, C, (B,)))
        _, dtxt = O.logits_bwd(dl, lctx)
        dprompts = O.text_encoder_bwd(sd, dtxt, tctx)
        O.scatter_prompt_grad(dprompts, layout, tuple(ctx.shape))
        t_txt = time.perf_counter() - t0
    step = t_img * (B / sample_images) + t_txt
    return {"value": round(B / step, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle/clip_oracle.py fp32 on {threads} host threads: image-tower forward on {sample_images} of {B} "
                      f"images ({t_img:.1f}s, extrapolated x{B // sample_images}) + full text tower fwd+bwd and head for "
                      f"{C} classes L={L} ({t_txt:.1f}s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--arch", default="ViT-B/16")
    ap.add_argument("--method", default="coop", choices=["coop", "vpt", "upt"])
    ap.add_argument("--cut", action="store_true", help="CUT_CONTEXTLEN text length instead of 77")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--trim-eot", action="store_true", help="evaluate the causal text tower only up to max(EOT) (exact; off by default)")
    ap.add_argument("--no-step-pipelining", action="store_true",
                    help="do not compute the next batch's image features underneath the current backward")
    ap.add_argument("--shard-text", action="store_true", help="class-shard the text tower over the ranks (many-class configs)")
    ap.add_argument("--grad-precision", default="split_grad", choices=["split_grad", "fast"],
                    help="split_grad (default): hi+lo operand pairs + fp32 attention in towers that carry a gradient (prompt "
                         "gradients within 1e-3 of the fp32 CPU path); fast: single 16-bit operands everywhere (~4e-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--all-kernel-timing", action="store_true", help="bracket every kernel class with marker events (slower)")
    args = ap.parse_args()

    from mvlpt_amd import distributed as D
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.trainer import MVLPT, SyntheticDataManager
    from mvlpt_amd.weights import ARCHS, make_state_dict

    rank, world, local = D.init_process_group()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    arch = ARCHS[args.arch]
    cfg = get_cfg_default()
    cfg.MODEL.BACKBONE.NAME = args.arch
    cfg.INPUT.SIZE = (arch.image_resolution, arch.image_resolution)
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE = args.batch
    cfg.TRAINER.MVLPT.COMPUTE_DTYPE = args.dtype
    cfg.TRAINER.MVLPT.GRAD_PRECISION = args.grad_precision
    cfg.TRAINER.CUT_CONTEXTLEN = args.cut
    n_ctx = n_vpt = 0
    if args.method in ("coop", "upt"):
        n_ctx = 16 if args.method == "coop" else 4
    if args.method in ("vpt", "upt"):
        n_vpt = 8 if args.method == "vpt" else 4
    cfg.TRAINER.MVLPT.COOP.N_CTX, cfg.TRAINER.MVLPT.VPT.N_CTX = n_ctx, n_vpt
    cfg.SEED = 1
    sd = make_state_dict(arch, seed=1)
    n_batches = 4
    dm = SyntheticDataManager(cfg, args.classes, n_batches, device=dev, seed=1234 + rank)
    trainer = MVLPT(cfg, dm=dm, clip_state_dict=sd)
    trainer.num_batches = 10 ** 9   # no LR-schedule step inside the timed region
    trainer.model.trim_text_to_eot = args.trim_eot
    if args.shard_text and world > 1:
        trainer.model.enable_class_sharding(rank, world)
    L_text = trainer.model.prompt_learner.tokenized_prompts.shape[1]
    eng = trainer.model.engine

    pipeline = not args.no_step_pipelining

    def step(i):
        nonlocal_pipeline = pipeline
        trainer.batch_idx = i
        nxt = dm.train_loader_x[(i + 1) % n_batches] if nonlocal_pipeline else None
        return trainer.forward_backward(dm.train_loader_x[i % n_batches], next_batch=nxt)

    # Multi-rank runs: make sure the cross-step prefetch is a win on this system before the timed region (it changes
    # how the gradient all-reduce interleaves with the side streams).  Untimed probe, 3 steps per mode, the decision is
    # the same on every rank (MAX over ranks of each mode's time).
    if pipeline and world > 1:
        def probe(flag):
            nonlocal pipeline
            pipeline = flag
            step(0)
            torch.cuda.synchronize(); D.barrier()
            t = time.perf_counter()
            for i in range(3):
                step(i)
            torch.cuda.synchronize()
            return D.all_reduce_max(time.perf_counter() - t, dev)
        t_pipe, t_plain = probe(True), probe(False)
        pipeline = t_pipe <= 1.15 * t_plain       # only a clear loss switches it off (3-step probes are noisy)
        if rank == 0 and not pipeline:
            print(f"note: cross-step prefetch disabled (probe: {t_pipe / 3 * 1e3:.2f} ms/step with, {t_plain / 3 * 1e3:.2f} without)",
                  file=sys.stderr)

    for i in range(args.warmup):
        out = step(i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    timing = not args.no_kernel_timing
    if timing:
        eng.profile_begin(all_kernels=args.all_kernel_timing)
    sample_every = 1 if args.all_kernel_timing else 4     # dispatch-timestamp timing costs ~2 us per launch: sample steps
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timing:
            eng.profile_pause(i % sample_every != 0)
        out = step(i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stats = eng.profile_end() if timing else {}
    elapsed = D.all_reduce_max(elapsed, dev)
    # Untimed extra pass: the same step with the two towers serialized on one stream, so each GEMM launch has the
    # chip to itself.  In the timed region above the text tower runs on a second stream underneath the image
    # tower; concurrent kernels stretch each other's durations, which inflates per-launch times (they then sum
    # to more than the step) without being slower overall.  Reported next to the timed-region figure.
    stats_serial = {}
    if timing and trainer.model.overlap_towers:      # every rank takes part (the step contains the gradient all-reduce)
        trainer.model.overlap_towers = False
        pipeline_saved, pipeline = pipeline, False       # no cross-step prefetch either: strictly one kernel at a time
        step(0)
        torch.cuda.synchronize()
        eng.profile_begin(all_kernels=False)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        stats_serial = eng.profile_end()
        trainer.model.overlap_towers = True
        pipeline = pipeline_saved
    loss = float(out["loss"])
    assert loss == loss, "loss is NaN"

    if rank == 0:
        B_global = args.batch * world
        ips = B_global * args.steps / elapsed
        L_alg = (trainer.model.prompt_learner.max_eot + 1) if args.trim_eot else L_text   # charge only evaluated positions
        gf_img = algorithmic_gflop_per_image(arch, B_global, args.classes, L_alg, n_ctx, n_vpt)
        line = {
            "metric": "prompt-tuning images/sec (fwd+bwd), ViT-B/16 bs=256, 1/2/4/8 MI355X",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (("BASELINE configs[1]: " if (args.method, args.arch, args.classes, args.batch) ==
                                     ("coop", "ViT-B/16", 100, 256) else "variant: ") +
                                    f"MVLPT {args.method} head, {args.arch}, {args.classes} classes, "
                                    f"n_ctx={n_ctx} n_vpt={n_vpt}, text L={L_text}, class token middle"),
                       "text_positions_evaluated": (trainer.model.prompt_learner.max_eot + 1) if args.trim_eot else L_text,
                       "step_pipelining": bool(pipeline and n_vpt == 0),
                       "grad_precision": args.grad_precision,
                       "per_gpu_batch": args.batch, "global_batch": B_global, "parallelism": f"dp{world}",
                       "text_tower": "class-sharded over ranks" if (args.shard_text and world > 1) else "replicated per GPU", "loss": round(loss, 5)},
            "step_mfma_fraction": round(ips / world * gf_img / (MFMA_PEAK_TFLOPS * 1e3), 4),
            "algorithmic_gflop_per_image": round(gf_img, 3),
        }
        traffic = None
        if os.path.isfile(TRAFFIC_FILE) and args.method == "coop" and args.batch == 256:
            with open(TRAFFIC_FILE) as f:
                traffic = json.load(f)       # {"bytes_per_launch": …, "source": "rocprofv3 --pmc …"} (offline PMC passes)
        n_sampled = len(range(0, args.steps, sample_every))
        if "gemm_bt" in stats:
            g = stats["gemm_bt"]
            # achieved = algorithmic FLOPs of the launches / time during which the kernel occupies the GPU.  The text
            # tower's GEMMs run on a second stream underneath the image tower's, so launch intervals overlap: the
            # occupied time is the UNION of the intervals (dispatch timestamps), not the sum of durations (which
            # would count concurrent stretches twice).  Both are reported; they coincide when nothing overlaps.
            tf = g["flops"] / (g["busy_ms"] * 1e-3) / 1e12
            tf_sum = g["flops"] / (g["ms"] * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_bt_kernel (all epilogues)", "achieved": round(tf, 1),
                                "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                                "traffic": traffic, "launches_per_step": g["launches"] // n_sampled, "steps_sampled": n_sampled,
                                "avg_launch_us": round(1e3 * g["ms"] / g["launches"], 2),
                                "busy_us_per_launch": round(1e3 * g["busy_ms"] / g["launches"], 2),
                                "achieved_sum_of_durations": round(tf_sum, 1),
                                "algorithmic_bytes_per_launch": int(g["bytes"] / g["launches"]),
                                "concurrency": "timed region: text tower on a 2nd stream and the next batch's image tower on a 3rd overlap; achieved = FLOPs / union of the launch intervals"}
            if "gemm_bt" in stats_serial:
                gs = stats_serial["gemm_bt"]
                tfs = gs["flops"] / (gs["ms"] * 1e-3) / 1e12
                line["roofline"]["serialized_towers"] = {"achieved": round(tfs, 1), "frac": round(tfs / MFMA_PEAK_TFLOPS, 4),
                                                         "avg_launch_us": round(1e3 * gs["ms"] / gs["launches"], 2),
                                                         "note": "3 untimed steps, everything on one stream, no cross-step prefetch"}
            line["kernel_ms_per_step"] = {k: round(v["ms"] / n_sampled, 3) for k, v in stats.items()}
        if world == 1 and not args.no_cpu_baseline and args.method == "coop":
            line["cpu_baseline"] = cpu_baseline_images_per_sec(arch, sd, args.batch, args.classes, L_text, n_ctx)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
