#!/usr/bin/env python
"""Benchmark of the VIRNet denoise forward (VIRAttResUNet, denoise-syn config) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one forward of the hot path over one resident batch of synthetic images.  Default workload = the configuration
BASELINE.json's metric is quoted on -- 256x256x3 images, 32 per GPU (configs[2]'s per-GPU shard; at N=8 the global batch is
configs[2]'s 256).  ``--size 128 --batch 64`` runs configs[1].  Scaling is weak: every rank owns ``--batch`` images, the only
collective is the start-up weight broadcast (RCCL), nothing is exchanged per image.

``--gpus N`` without a torchrun environment re-executes itself through ``torch.distributed.run`` (N ranks on this node), so the
plain command works as well as the torchrun one.

``--dry-run-topology`` (with ``--gpus N``): no timed steps -- every rank reports its device UUID, the run fails if two ranks share a device
although enough are visible (virnet_amd.dist.rank_topology), and the start-up weight broadcast is timed on its own.  ``--guard sync|deferred``:
how the timed inference forwards learn of an fp16-range overflow (engine.guard_check_mode; default sync = the product's default: one flag
read per forward; round 5's lines ran deferred -- the line carries BOTH modes' values, `summary.guard`, measured 1 584 / 1 570 img/s on one box).  Defaults: 30 timed steps behind 10 warm-up steps (the socket needs ~0.2 s of load to settle at its cap).

Rank 0 prints ONE JSON line whose FIRST key, ``summary``, holds the seven headline values; besides the contract fields it carries
  roofline     : the dominant kernel = the launch group of the C->C 3x3 res-block convs with the most time (conv_f16_kernel by
                 default; conv_wino_row_kernel / conv_mfma_kernel with VIRNET_CONV_FORM=wino|direct).  Launch durations are
                 measured with HIP events recorded on the launch stream around every launch inside the timed region (rank 0).
                 `achieved` = FLOPs the kernel EXECUTES on its matrix pipe per second (algorithmic 2*MAC x the form's factor:
                 f16x3 3 products per MAC, Winograd 16/36, direct 1), `peak` = that pipe's dense peak (f16 2500, fp32 157.3
                 TFLOP/s), so `frac` <= 1 is the pipe's utilisation; `algorithmic_tflops` is the contract's 2*MAC rate.
  cpu_baseline : the CPU oracle (oracle/cpu_ref.py, kind "port") timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from virnet_amd import dist as vdist  # noqa: E402
from virnet_amd import engine  # noqa: E402
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks import VIRAttResUNet, VIRAttResUNetSR  # noqa: E402
from virnet_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402

# scripts/denoising_virnet_syn.py:62-71
SYN_CFG = dict(n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input", noise_avg=False)
# scripts/sisr_virnet_syn.py:53-63 / scripts/testing_demo.py:52-63
SISR_CFG = dict(n_feat=[96, 160, 224], dep_S=5, dep_K=8, n_resblocks=2, noise_cond=True, kernel_cond=True, extra_mode="Both",
                noise_avg=True)
SISR_GFLOP_PER_IMAGE = 180.159         # x4, LR 64x64 -> 256x256 (RNet 178.92, SNet 0.925, KNet 0.311)
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0          # same guide: BF16/F16 MFMA ~2.5 PFLOP/s dense (16x the fp32 matrix rate)
# matrix-pipe FLOPs executed per algorithmic FLOP (2*MAC of the direct 3x3 convolution), and the pipe they run on
FORMS = {"wx4": (1.5, F16_MFMA_PEAK_TFLOPS, "Winograd F(4,3) along x (18 instead of 36 k-steps per 4 output pixels) with split-fp16 position products: 1.5 executed "
                                            "FLOP per algorithmic FLOP on v_mfma_f32_32x32x16_f16, fp32 accumulation; layers / shapes it does not cover run as f16x3"),
         "f16x3": (3.0, F16_MFMA_PEAK_TFLOPS, "split-fp16 operands: 3 products per MAC on v_mfma_f32_32x32x16_f16, fp32 accumulation"),
         "bf16": (1.0, F16_MFMA_PEAK_TFLOPS, "bf16-rounded operands: 1 product per MAC on v_mfma_f32_32x32x16_bf16, fp32 accumulation (C->C 3x3 convs: "
                                            "forward, input gradients AND weight gradients; thin layers, strided / transposed convs split-fp16 or fp32)"),
         "wino": (16.0 / 36.0, FP32_MFMA_PEAK_TFLOPS, "Winograd F(2x2,3x3), fp32: 16/36 of the algorithmic MACs on v_mfma_f32_32x32x2_f32"),
         "direct": (1.0, FP32_MFMA_PEAK_TFLOPS, "direct implicit GEMM, fp32 on v_mfma_f32_32x32x2_f32")}
KFLOP_PER_PIXEL = 4988.736             # conv FLOPs (2*MAC) of the denoise-syn forward per padded pixel


def build_net(device, task="denoise"):
    net = VIRAttResUNet(im_chn=3, sigma_chn=1, **SYN_CFG) if task == "denoise" else VIRAttResUNetSR(im_chn=3, sigma_chn=1, kernel_chn=3, **SISR_CFG)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    return net, sd


def cpu_baseline(sd, size: int, budget_s: float = 60.0):
    """Time the CPU oracle on the host cores on a bounded sample of the same workload (N = 4 @256^2, N = 8 @128^2).

    torch's CPU convs stop scaling (and then collapse) long before 256 threads on these small batches, so a few thread counts are
    tried (one warm-up + one timed run each) and the fastest is then timed until there are THREE runs of it: `value` is their median,
    `cores` the threads used.  The one-thread figure (single image) is reported beside it; the all-cores run (0.13 images/s on the
    256-core host: a minute per run) is in BASELINE.md 3 and no longer part of the default line."""
    from oracle import cpu_ref
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nimg = 4 if size >= 256 else 8
    x = synth_images(nimg, 3, size, size)

    def run_once(xx=x):
        t0 = time.perf_counter()
        cpu_ref.virnet_denoise(sd, xx, **SYN_CFG)
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    best_t, best_n, tried = None, None, {}
    with torch.no_grad():
        for n in sorted({min(avail, k) for k in (8, 16, 32, 64)}):
            if tried and time.perf_counter() - t_begin > budget_s * 0.5:
                break
            torch.set_num_threads(n)
            run_once()                                   # warm-up (thread pool, oneDNN primitive cache)
            t = run_once()
            tried[n] = round(nimg / t, 3)
            if best_t is None or t < best_t:
                best_t, best_n = t, n
        torch.set_num_threads(best_n)
        run_once()                                       # (the pool was resized: warm it again)
        times = [run_once() for _ in range(3)]
        torch.set_num_threads(1)
        x1 = x[:1]
        run_once(x1)
        t1 = run_once(x1)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(nimg / med, 3), "unit": "images/s", "cores": best_n, "kind": "port",
            "runs_s": [round(t, 3) for t in times],
            "one_thread": {"threads": 1, "images_per_s": round(1.0 / t1, 4), "sample": f"[1,3,{size},{size}]"},
            "by_threads": tried, "cores_available": avail,
            "sample": f"oracle/cpu_ref.virnet_denoise on [{nimg},3,{size},{size}] fp32, torch CPU, {best_n} threads "
                      f"(fastest of the thread counts tried; {avail} cores available), median of {len(times)} runs"}


class PowerSampler:
    """Socket power and shader clock of ONE device while a timed region runs: a thread reads the device's hwmon files
    (power1_input in uW, freq1_input = sclk in Hz) every `period` seconds.  The dominant kernels of this path run the socket at its
    power cap (profiles/r04_probes.md): the line reports it so that a roofline fraction can be read against the right ceiling."""

    def __init__(self, dev_index: int, period: float = 0.02):
        import glob
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.dir, self.cap_w = None, None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            cands = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if cands and os.path.exists(os.path.join(cands[0], "power1_input")):
                self.dir = cands[0]
                with open(os.path.join(self.dir, "power1_cap")) as f:
                    self.cap_w = int(f.read()) / 1e6
        except Exception:          # noqa: BLE001  (no sysfs access: the block is simply absent)
            self.dir = None
        self._thread = threading.Thread(target=self._run, daemon=True) if self.dir else None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return int(f.read())

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append((time.perf_counter(), self._read("power1_input") / 1e6, self._read("freq1_input") / 1e6))
            except Exception:      # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self._thread:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread:
            self._thread.join()
        return False

    def summary(self, skip_s: float = 0.15):
        """mean / max over the samples of the region, the first `skip_s` seconds (the firmware's averaging window) left out"""
        if not self.samples:
            return None
        t0 = self.samples[0][0]
        body = [s for s in self.samples if s[0] - t0 >= skip_s] or self.samples
        pw, ck = [s[1] for s in body], [s[2] for s in body]
        return {"socket_w_mean": round(sum(pw) / len(pw), 1), "socket_w_max": round(max(pw), 1), "cap_w": self.cap_w,
                "sclk_mhz_mean": round(sum(ck) / len(ck), 1), "sclk_mhz_min": round(min(ck), 1), "samples": len(body),
                "source": "hwmon power1_input / freq1_input of the device, sampled every %d ms inside the timed region" % int(self.period * 1e3)}


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def summary(self):
        return None


def run_config(extra, timeout=420):
    """One BASELINE config as a child run of this script (own process: the conv form of the bf16 variant is process state); returns the
    reduced JSON line or an error string."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-configs"] + extra
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "timeout", "cmd": " ".join(extra)}
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": (out.stderr or out.stdout)[-400:], "cmd": " ".join(extra)}
    d = json.loads(lines[-1])
    r = d.get("roofline") or {}
    return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
            "dtype": d["dtype"], "workload": d["config"]["workload"], "cmd": "bench.py " + " ".join(extra),
            "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "avg_launch_ms",
                                               "launches_per_step", "share_of_conv_time", "groups")} if r else None,
            "power": d.get("power")}


def load_pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/), else None."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps before the timed region (default 10: the socket needs ~0.2 s of load to settle at its cap / clock -- with 3 the first timed steps still run on the ramp: 1 541 vs 1 571 img/s in one process, profiles/r05_bench_default_line.json)")
    ap.add_argument("--size", type=int, default=256, help="image height=width")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 32 @256, 64 @128)")
    ap.add_argument("--task", default="denoise", choices=["denoise", "sisr", "train", "train_sisr"],
                    help="sisr = BASELINE configs[3]: VIRAttResUNetSR x4 on LR 64x64 (-> 256x256), 16 images per GPU; "
                         "train = configs[4]: denoise-syn forward + ELBO + backward (+Adam with --optimizer) on 128x128, 32 per GPU, fp32; "
                         "train_sisr = the SISR step (train_SISR.py:207-224) on configs[3]'s shape: forward x4 + elbo_sisr + backward")
    ap.add_argument("--optimizer", action="store_true", help="train task: include grad clipping + Adam step in the timed step")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="train task: bf16 = BASELINE configs[4]'s variant -- the C->C 3x3 convs of forward and backward (input AND weight "
                         "gradients) run with bf16-rounded operands, one product per MAC, fp32 accumulation; all other layers stay fp32-class")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="default N=1 denoise run: skip the `steady_state` re-measurement and the `configs` "
                    "block (BASELINE configs[1], [3], [4] timed by child runs of this script)")
    ap.add_argument("--dry-run-topology", action="store_true", help="multi-GPU pre-flight (no timed steps): every rank reports its device UUID, "
                    "the run FAILS if two ranks share a device although enough devices are visible, the start-up weight broadcast is timed on its own")
    ap.add_argument("--guard", default="sync", choices=["sync", "deferred"], help="range-guard check of the timed forwards (engine.guard_check_mode): "
                    "sync (default since round 6: the product's default -- one flag read at the end of every forward) or deferred (round 5's headline mode: "
                    "no host wait per forward, outputs NaN-poisoned on overflow, every forward checked by guard_poll before the clock stops); the OTHER mode is "
                    "timed right behind the headline region and reported in `other_guard_mode` / `summary.guard`")
    args = ap.parse_args()
    sisr = args.task in ("sisr", "train_sisr")
    training = args.task in ("train", "train_sisr")
    train_sisr = args.task == "train_sisr"
    if args.dtype == "bf16":
        if not training:
            raise SystemExit("--dtype bf16 is the training variant (inference must be fp32-class: BASELINE.md, 7.5e-3 error at bf16)")
        os.environ["VIRNET_CONV_FORM"] = "bf16"
    if sisr and args.size == 256:
        args.size = 64                      # LR size; the output is 4x
    if training and not sisr and args.size == 256:
        args.size = 128                     # configs/denoising_syn.json:6 patch_size
    batch = args.batch if args.batch is not None else (16 if sisr else 32 if (training or args.size >= 256) else 64)

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU over RCCL, train_denoising_syn.py:280-297's mp.spawn)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank, local_rank, world = vdist.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the product path has no CPU fallback")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())   # (modulo: lets a 1-GPU box rehearse N ranks over gloo)
    torch.cuda.set_device(dev)
    if world > 1:
        # N Python processes share the host: cap every rank's CPU thread pool (torch's default is one thread per core, i.e. N x 256
        # runnable threads on the 8-GPU node) -- the hot path enqueues kernels from ONE thread, the pools only serve host-side glue
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(8, cores // world)))

    net, sd = build_net(dev, "sisr" if sisr else "denoise")
    fwd = (lambda t: net(t, 4)) if sisr else net
    if rank == 0:
        net.load_state_dict(sd, strict=True)       # other ranks keep their random init until the broadcast
    net = net.to(dev).eval()
    t0 = time.perf_counter()
    bcast_bytes = vdist.broadcast_parameters(net, src=0)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3

    topo = None
    if world > 1:
        # Every rank on its OWN device: local_rank % device_count lets a 1-GPU box rehearse N ranks (tests/test_dist_gpu.py), but a real
        # N-GPU run whose ranks pile up on one device (a launcher that hides devices per rank, a wrong LOCAL_RANK) must not produce a
        # "scaling" number -- it fails here.  Reference launch pattern: train_denoising_syn.py:280-297 (one process per visible GPU).
        props = torch.cuda.get_device_properties(dev)
        mine = (str(getattr(props, "uuid", "")) or f"index{dev.index}", torch.cuda.device_count())
        seen = [None] * world
        torch.distributed.all_gather_object(seen, mine)
        try:
            topo = vdist.rank_topology(seen, world)
        except RuntimeError as exc:
            raise SystemExit(f"bench.py: {exc}")
    if args.dry_run_topology:
        reps = []
        for _ in range(3):
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            t1 = time.perf_counter()
            nb = vdist.broadcast_parameters(net, src=0)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t1) * 1e3)
        reps = [vdist.max_over_ranks(r, dev) for r in reps]
        if rank == 0:
            print(json.dumps({"dry_run_topology": dict(topo or {"ranks": 1, "ranks_seen": 1}, backend=(torch.distributed.get_backend() if world > 1 else None),
                                                       broadcast_bytes=nb, broadcast_ms=[round(r, 3) for r in reps],
                                                       broadcast_gb_per_s=round(nb / (min(reps) * 1e-3) / 1e9, 2) if nb else None,
                                                       first_broadcast_ms_incl_init=round(bcast_ms, 3))}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    # this rank's shard of the global batch (weak scaling: `batch` images per rank), resident in HBM before timing
    a, b = vdist.shard_range(batch * world, world, rank)
    x = synth_images(b - a, 3, args.size, args.size, seed=20240916 + rank).to(dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    if train_sisr:
        # one step of train_SISR.py:207-224 on resident synthetic data: forward (x4), elbo_sisr (host-side PyTorch, ELBO_simple.py:82-138),
        # backward through the per-conv HIP kernels; optionally the three clip_grad_norm_ + Adam
        from virnet_amd.loss import elbo_sisr
        net.train()
        nloc = b - a
        im_hr = synth_images(nloc, 3, args.size * 4, args.size * 4, seed=7 + rank).to(dev)
        kinfo_gt = torch.tensor([[1.2, 0.8, 0.1]], device=dev).repeat(nloc, 1)
        nlevel = torch.full((nloc, 1, 1, 1), 2e-3, device=dev)
        alpha0 = 0.5 * torch.tensor([9.0 ** 2], device=dev)
        kappa0 = torch.tensor([50.0], device=dev)
        opt = torch.optim.Adam(net.parameters(), lr=2e-4) if args.optimizer else None
        groups = {key: [p for nm, p in net.named_parameters() if key in nm.lower()] for key in ("rnet", "snet", "knet")}

        # N > 1: data-parallel training as train_SISR.py (DDP): the per-layer autograd nodes of the SISR step fire torch DDP's hooks
        # layer by layer, so the bucketed RCCL all-reduces overlap with the rest of the backward
        model_sr = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index]) if world > 1 else net

        def fwd(t):
            for p in net.parameters():
                p.grad = None
            mu_, kinfo_, sig_ = model_sr(t, 4)
            loss = elbo_sisr(mu=mu_, sigma_est=sig_, kinfo_est=kinfo_, im_hr=im_hr, im_lr=t, sigma_prior=nlevel, alpha0=alpha0, kinfo_gt=kinfo_gt,
                             kappa0=kappa0, r2=1e-4, eps2=1e-5, sf=4, k_size=21, penalty_K=[0.02, 2], shift=False, downsampler="Bicubic")[0]
            loss.backward()
            if opt is not None:
                torch.nn.utils.clip_grad_norm_(groups["rnet"], 5e2)
                torch.nn.utils.clip_grad_norm_(groups["snet"], 1e2)
                torch.nn.utils.clip_grad_norm_(groups["knet"], 5e2)
                opt.step()
            return (mu_.detach(),)
    elif training:
        # one step of train_denoising_syn.py:171-184 on resident synthetic data: forward, ELBO (host-side PyTorch,
        # loss/ELBO_simple.py:23-53), backward through the HIP kernels, optionally clip + Adam
        net.train()
        gt = synth_images(b - a, 3, args.size, args.size, seed=7 + rank).to(dev)
        sigma_gt = (0.02 + 0.25 * synth_images(b - a, 1, args.size, args.size, seed=11 + rank).to(dev)) ** 2
        alpha0 = torch.tensor([0.5 * 7 ** 2], dtype=torch.float32, device=dev)      # var_window 7 (configs/denoising_syn.json:38)
        beta0, eps2 = alpha0 * sigma_gt, 1e-6
        opt = torch.optim.Adam(net.parameters(), lr=2e-4) if args.optimizer else None
        p_r = [p for n_, p in net.named_parameters() if "rnet" in n_.lower()]
        p_s = [p for n_, p in net.named_parameters() if "snet" in n_.lower()]

        from virnet_amd.loss import elbo_denoising_simple

        # N > 1: data-parallel training as train_denoising_syn.py:71 (DDP) -- gradients averaged over the ranks every step, the
        # bucketed RCCL all-reduces started from inside the backward (virnet_amd/dist.py)
        model = vdist.DistributedTrainer(net) if world > 1 else net

        def fwd(t):
            for p in net.parameters():
                p.grad = None
            mu_, sig_ = model(t)
            loss = elbo_denoising_simple(mu_, sig_, t, gt, eps2, alpha0, beta0)[0]
            loss.backward()
            if opt is not None:
                torch.nn.utils.clip_grad_norm_(p_r, 1e3)
                torch.nn.utils.clip_grad_norm_(p_s, 1e2)
                opt.step()
            return (mu_.detach(),)

    if not training:
        os.environ["VIRNET_GUARD_CHECK"] = args.guard      # (inference forwards; the training step's forward is one autograd Function with its own check)
    with torch.set_grad_enabled(training):
        # (everything the timed region needs is built BEFORE the warm-up steps: the sampler's sysfs look-ups took tens of milliseconds between
        # the last warm-up step and the first timed one -- long enough for the socket to leave its clock, and the first timed steps then ran on
        # the ramp again: 1 366 W / 1 524 img/s in the contract's region against 1 391 W / 1 559 img/s in the region timed right behind it)
        timer = ops.LaunchTimer() if (rank == 0 and not args.no_roofline) else None
        psamp = PowerSampler(dev.index)                       # (every rank samples ITS device: a throttling rank must be visible in `multi_gpu.per_rank`)
        for _ in range(args.warmup):
            fwd(x)
        engine.guard_poll()
        torch.cuda.synchronize()
        barrier()
        ops.set_launch_timer(timer)     # two event records per conv launch, on the launch stream (~us of host time each)
        with psamp:
            t0 = time.perf_counter()
            c0 = time.thread_time()
            for _ in range(args.steps):
                mu = fwd(x)[0]
            host_enqueue = time.perf_counter() - t0          # wall time until the K steps are enqueued (the device runs behind; a full
            host_cpu = time.thread_time() - c0               # launch queue blocks here) and the CPU time this thread spent doing it
            torch.cuda.synchronize()
            engine.guard_poll()                              # deferred guard: every forward of the region has been checked when the clock stops
            barrier()
            elapsed = time.perf_counter() - t0
        ops.set_launch_timer(None)
        power = psamp.summary()
        # what ONE step costs the host with an empty launch queue (all ranks at once: N processes share the host's cores) -- the
        # number a rank's device time per step must exceed for the rank not to be host-bound
        # (measured with the range guard's end-of-forward flag read switched off for these three steps: the read waits for the device)
        one = []
        guard_env = os.environ.get("VIRNET_RANGE_GUARD")
        os.environ["VIRNET_RANGE_GUARD"] = "0"
        try:
            for _ in range(3):
                torch.cuda.synchronize()
                barrier()
                t1 = time.perf_counter()
                fwd(x)
                one.append(time.perf_counter() - t1)
        finally:
            if guard_env is None:
                os.environ.pop("VIRNET_RANGE_GUARD", None)
            else:
                os.environ["VIRNET_RANGE_GUARD"] = guard_env
        torch.cuda.synchronize()
        host_one_step = sorted(one)[1]
        # The headline's guard mode is --guard (default sync = the product default, engine.guard_check_mode); the OTHER mode (deferred: the mode of
        # pipelined callers, round 5's headline) is timed right behind it on the same box, same K steps, so that the line carries both
        other_guard = None
        if not training and world == 1 and not args.no_configs:
            og = "sync" if args.guard == "deferred" else "deferred"
            os.environ["VIRNET_GUARD_CHECK"] = og
            try:
                for _ in range(3):
                    fwd(x)
                engine.guard_poll()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    mu = fwd(x)[0]
                torch.cuda.synchronize()
                engine.guard_poll()
                el_o = time.perf_counter() - t1
            finally:
                os.environ["VIRNET_GUARD_CHECK"] = args.guard
            other_guard = {"mode": og, "value": round(batch * args.steps / el_o, 2), "ms_per_step": round(el_o / args.steps * 1e3, 3), "steps": args.steps,
                           "note": f"the same K steps with VIRNET_GUARD_CHECK={og}, timed right behind the headline region (which ran with {args.guard})"}
        # A second, longer region (the contract's K steps are ~0.4 s: short against the firmware's power averaging and the box-to-box
        # spread): >= 50 steps when that stays under ~5 s.  Reported beside `value`, never instead of it.
        steady = None
        main_default = (not training and not sisr and world == 1 and not args.no_configs)
        if main_default and elapsed / args.steps * 50 <= 5.0:
            n2 = max(50, args.steps)
            with PowerSampler(dev.index) as ps2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n2):
                    mu = fwd(x)[0]
                torch.cuda.synchronize()
                engine.guard_poll()
                el2 = time.perf_counter() - t0
            steady = {"steps": n2, "ms_per_step": round(el2 / n2 * 1e3, 3), "value": round(batch * n2 / el2, 2), "unit": "images/s",
                      "power": ps2.summary()}
    elapsed_local = elapsed
    elapsed = vdist.max_over_ranks(elapsed, dev)
    assert torch.isfinite(mu).all()
    # per-rank diagnostics of a multi-GPU run (one all-gather of small python objects, outside the timed region): every rank's own
    # rate, the device it ran on (distinct UUIDs = really N GPUs), the collective library
    diag = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device_index": dev.index, "uuid": str(getattr(props, "uuid", "")), "name": props.name,
                "images_per_s": round((b - a) * args.steps / elapsed_local, 2), "seconds": round(elapsed_local, 4),
                "host_ms_one_step_empty_queue": round(host_one_step * 1e3, 3),
                "enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3), "host_cpu_ms_per_step": round(host_cpu / args.steps * 1e3, 3),
                "cpu_threads": torch.get_num_threads(),
                "power": ({k: power.get(k) for k in ("socket_w_mean", "socket_w_max", "cap_w", "sclk_mhz_mean", "sclk_mhz_min", "samples")} if power else None)}
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        try:
            nccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:          # (gloo rehearsal on a box without RCCL)
            nccl = None
        diag = {"backend": torch.distributed.get_backend(), "rccl_version": nccl, "ranks_seen": len({g["uuid"] or g["device_index"] for g in gathered}),
                "per_rank": gathered, "broadcast_bytes": bcast_bytes, "broadcast_ms": round(bcast_ms, 3)}

    roof = None
    if timer is not None:
        summ = timer.summary()
        def kname(k):
            if k[0] == "thin":
                return "conv3x3_thin<cout=%d>" % k[3]
            if k[0] == "wgrad":
                return "conv_wgrad<ks=%d,s=%d,t=%d>" % k[1:]
            if k[0] == "wgrad_f16":
                return "chsplit x2 + conv_wgrad_f16 + reduce <ks=%d,s=%d,t=%d>" % k[1:]
            if k[0] == "wgrad_f16_s2":
                return "chsplit_s2 + chsplit + conv_wgrad_f16<S=2> + reduce <ks=%d,s=%d,t=%d>" % k[1:]
            if k[0] == "exit":
                return "conv_exit<cout=%d>" % k[1]
            if k[0] == "entry":
                return "conv_entry<cout=%d>" % k[1]
            if k[0] in ("wino", "f16x3", "f16x3_s2", "f16x3_t", "bf16", "wx4"):
                return "conv_%s<cout=%d>" % ({"f16x3": "f16", "f16x3_s2": "f16_s2", "f16x3_t": "f16_pw(convT)", "wino": "wino", "bf16": "bf16", "wx4": "wx4"}[k[0]], k[1])
            return "conv_mfma<%d,%d,%d,%d>" % k
        # dominant kernel = the launch group of the stride-1 3x3 res-block convs with the most time
        cands = ([k for k in summ if k[0] in ("wino", "f16x3", "bf16", "wx4")] or [k for k in summ if k[0] == 3 and k[1] == 1 and k[3] == 3]
                 or [k for k in summ if k[0] == 3 and k[1] == 1])
        dom = max(cands, key=lambda k: summ[k]["ms"]) if cands else None
        d = summ.get(dom)
        if d:
            form = dom[0] if dom[0] in FORMS else "direct"
            factor, peak, how = FORMS[form]
            avg_ms = d["ms"] / d["launches"]
            algorithmic = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
            total_ms = sum(v["ms"] for v in summ.values())
            pmc = load_pmc_traffic() if (not sisr and not training and args.size == 256 and batch == 32) else None   # measured on this workload only
            kern = {"wx4": "conv_wx4_kernel<NREP,EPI,PRE> (3x3 stride-1, %d channels)" % dom[1],
                    "f16x3": "conv_f16_kernel<MREP,NREP,EPI> (3x3 stride-1, %d channels)" % dom[1],
                    "bf16": "conv_f16_kernel<MREP,NREP,EPI,BF=1> (3x3 stride-1, %d channels, bf16 operands)" % dom[1],
                    "wino": "conv_wino_row_kernel<G,false,WPU> (3x3 stride-1, %d channels)" % dom[1]}.get(form) or "conv_mfma_kernel<%d,%d,%d,%d>" % dom
            # `frac` prices the FLOPs the kernel EXECUTES on its pipe (emulation products included); `frac_algorithmic` the contract's
            # 2*MAC of the direct convolution.  `traffic` is not measured inside a timed run (PMC passes serialise the kernels): it is the
            # HBM bytes per launch of this kernel from the committed rocprofv3 --pmc pass named in `traffic_source`, or null.
            roof = {"bound": "mfma", "achieved": round(algorithmic * factor, 2), "peak": peak, "unit": "TFLOP/s",
                    "peak_of": ("f16 MFMA dense (v_mfma_f32_32x32x16_f16, 2.5 PFLOP/s nominal)" if peak == F16_MFMA_PEAK_TFLOPS
                                else "fp32 MFMA dense (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s)"),
                    "frac": round(algorithmic * factor / peak, 4), "frac_algorithmic": round(algorithmic / peak, 4),
                    "traffic": (pmc or {}).get("hbm_bytes_per_launch") if (pmc or {}).get("form", "wino") == form else None,
                    "traffic_measured_in_run": False,
                    "traffic_source": ("profiles/pmc_latest.json (%s)" % (pmc or {}).get("source", "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes")
                                       if (pmc or {}).get("form", "wino") == form else None),
                    "kernel": kern, "algorithm": how,
                    "ceiling_note": ("the launch group runs the socket at its package power cap (see `power`: mean W of the timed region against cap_w; "
                                     "profiles/r04_probes.md: 1399-1400 W of 1400 W on this kernel alone, 2.4 GHz and 16 % less time on all-zero operands) -- "
                                     "frac is bounded by joules per MFMA on this data, not by issue slots"),
                    "algorithmic_tflops": round(algorithmic, 2), "executed_per_algorithmic_flop": round(factor, 4),
                    "algorithmic_over_fp32_mfma_peak": round(algorithmic / FP32_MFMA_PEAK_TFLOPS, 4),
                    "launches_per_step": d["launches"] // args.steps,
                    "avg_launch_ms": round(avg_ms, 4), "flop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
                    "flop_unit": "GFLOP (2*MAC of the direct 3x3 convolution, algorithmic: SURVEY.md 8d)",
                    "share_of_conv_time": round(d["ms"] / total_ms, 4),
                    # the launch groups of this workload by time (a group = every launch of one kernel family at one output-channel count, whatever
                    # its epilogue / pre-activation instantiation): `kernel` above is groups[0] among the stride-1 3x3 convs
                    "groups": [{"group": kname(k), "ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["launches"] // args.steps,
                                "share_of_conv_time": round(v["ms"] / total_ms, 4),
                                "frac_algorithmic": (round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / FORMS[k[0]][1], 4) if k[0] in FORMS and v["ms"] > 0 else None)}
                               for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:4]],
                    "by_kernel_ms_per_step": {kname(k): round(v["ms"] / args.steps, 3) for k, v in sorted(summ.items(), key=lambda kv: str(kv[0]))}}
    if world > 1:
        torch.distributed.barrier()

    if rank == 0:
        imgs = batch * world * args.steps
        value = imgs / elapsed
        hp = (args.size + 3) // 4 * 4
        gflop_img = SISR_GFLOP_PER_IMAGE * (args.size / 64.0) ** 2 if sisr else KFLOP_PER_PIXEL * hp * hp / 1e6
        if training:
            gflop_img *= 3.0                  # forward + input-gradient + weight-gradient convs (~3x forward)
        out = {
            "summary": None,               # (first key on purpose: the driver keeps the head of the line -- filled in below)
            "metric": (f"images/sec (SISR x4 training step fwd+ELBO+bwd{'+Adam' if args.optimizer else ''}, LR {args.size}x{args.size}x3 -> {4 * args.size}x{4 * args.size})" if train_sisr else
                       f"images/sec (denoise-syn training step fwd+ELBO+bwd{'+Adam' if args.optimizer else ''}, {args.size}x{args.size}x3)" if training else
                       f"images/sec (SISR x4 fwd, LR {args.size}x{args.size}x3 -> {4 * args.size}x{4 * args.size})" if sisr else
                       "images/sec (256x256x3 denoise fwd)" if args.size == 256 else f"images/sec ({args.size}x{args.size}x3 denoise fwd)"),
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 operands, f32 accumulate" if args.dtype == "bf16" else "f32 (3x3 conv products as split-fp16 MFMA pairs, f32 accumulate)"),
            "data": "synthetic",
            "config": {"workload": (f"VIRAttResUNetSR x4 forward (n_feat 96/160/224, 2 res-blocks, dep_S 5, dep_K 8, extra_mode Both), LR {args.size}x{args.size}x3 "
                                    if sisr else f"VIRAttResUNet denoise-syn forward (n_feat 96/192/288, 3 res-blocks, dep_S 5), {args.size}x{args.size}x3 ")
                                   + f"U[0,1) images, {batch} per GPU per step (global batch {batch * world}), random-init weights, inputs resident in HBM",
                       "images_per_gpu": batch, "global_batch": batch * world, "image": [3, args.size, args.size],
                       "arithmetic": "fp32 tensors and accumulation; C->C 3x3 convs: " + FORMS[ops.conv_form()][2],
                       "range_guard": (None if training else f"{args.guard} (engine.guard_check_mode; every timed forward is checked before the clock stops)"),
                       "parallelism": (f"image-sharded x{world}, one weight broadcast ({bcast_bytes} B, {bcast_ms:.1f} ms incl. sync), "
                                       + ("gradient all-reduce per step (fp32 buckets, started inside the backward)" if (training and world > 1)
                                          else "no per-image collective"))},
            "whole_net": {"gflop_per_image": round(gflop_img, 3), "achieved_tflops_per_gpu": round(value / world * gflop_img / 1e3, 2),
                          "frac_of_fp32_mfma_peak": round(value / world * gflop_img / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)},
            "roofline": roof,
            "power": power,
            "steady_state": steady,
            "other_guard_mode": other_guard,
            "multi_gpu": diag,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1 or sisr or training) else cpu_baseline(sd, args.size),
        }
        if main_default and args.size == 256:
            # every other BASELINE config in the same line: >= 10 timed steps each, own dominant-kernel roofline
            torch.cuda.empty_cache()
            out["configs"] = {
                "configs[1] denoise fwd 128x128 x64": run_config(["--size", "128", "--batch", "64", "--steps", "20", "--warmup", "5"]),
                "configs[3] SISR x4 fwd, LR 64x64 x16": run_config(["--task", "sisr", "--steps", "20", "--warmup", "5"]),
                "configs[4] train fwd+ELBO+bwd 128x128 x32, bf16 (as written)": run_config(["--task", "train", "--dtype", "bf16", "--steps", "10", "--warmup", "3"]),
                "configs[4] train, fp32-class arithmetic": run_config(["--task", "train", "--steps", "10", "--warmup", "3"]),
                "SISR x4 training step, LR 64x64 x16": run_config(["--task", "train_sisr", "--steps", "10", "--warmup", "3"]),
                # SURVEY 8(d) words the metric as a GLOBAL batch of 256 at every N: the N=1 spot value of that wording (all 256 images on this GPU)
                "configs[2] global batch 256 on ONE GPU (256x256 x256)": run_config(["--size", "256", "--batch", "256", "--steps", "3", "--warmup", "1"]),
            }
            cf = list(out["configs"].values())
            val = lambda d: d.get("value") if isinstance(d, dict) else None        # noqa: E731
            out["summary"] = {"unit": "images/s, one MI355X", "fwd256_x32": out["value"], "fwd128_x64": val(cf[0]), "sisr_x4_x16": val(cf[1]),
                              "train_bf16": val(cf[2]), "train_f32class": val(cf[3]), "sisr_train": val(cf[4]), "fwd256_x256_global": val(cf[5]),
                              "roofline_frac_algorithmic": (roof or {}).get("frac_algorithmic"), "socket_w_mean": (power or {}).get("socket_w_mean"),
                              "guard": {args.guard: out["value"], (other_guard or {}).get("mode", "other"): (other_guard or {}).get("value"),
                                        "headline_mode": args.guard, "product_default": "sync"}}
        if out["summary"] is None:
            out["summary"] = {"unit": out["unit"], "value": out["value"], "n_gpus": world, "roofline_frac_algorithmic": (roof or {}).get("frac_algorithmic")}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
