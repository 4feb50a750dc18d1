#!/usr/bin/env python
"""Energy per launch (socket W x ms = J) of ONE conv launch repeated back to back -- the unit VERDICT r04 asks every A/B of the
power-capped launches to be judged in.

    python tools/probes/joule_ledger.py --shape l0 --mode pre [--seconds 3] [--tag NAME]           one build (VIRNET_HIP_LIB selects it)
    python tools/probes/joule_ledger.py --sweep base,nomfma,norda,... [--shapes l0] [--modes pre,res] [--repeat 2]
        runs itself once per ledger build virnet_amd/lib/libvirnet_hip_led_<name>.so (tools/build_ledger.sh) and prints one table row each

A row: ms per launch (wall time of the loop / launches), mean socket power and shader clock while the loop runs (hwmon power1_input /
freq1_input of the device, every 10 ms, first 0.3 s dropped), J = W x ms, and J_dyn = (W - idle W) x ms with the idle power sampled in
the same process before the first launch.  `--env K=V,...` sets library knobs (VIRNET_WX4_ROWS=8 ...) for an A/B of shipped forms."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def one(a):
    import torch
    from bench import PowerSampler
    from bench_conv import SHAPES
    from virnet_amd import ops
    from virnet_amd.networks.params import ConvParam
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    with PowerSampler(0, period=0.01) as ps:          # idle power of this box, GPU context alive, nothing queued
        time.sleep(0.8)
    idle = ps.summary(skip_s=0.2)
    torch.manual_seed(0)
    if a.op == "conv":
        n, h, w, c = SHAPES[a.shape]
        cp = ConvParam(c, c, 3).cuda()
        x = torch.rand(n, h, w, c, device=dev) - 0.5
        res = torch.rand(n, h, w, c, device=dev) - 0.5
        z = os.environ.get("BENCH_ZEROS", "0")
        if z in ("1", "x"):
            x.zero_(); res.zero_()
        if z in ("1", "w"):
            with torch.no_grad():
                cp.weight.zero_(); cp.bias.zero_()
        kw = {"res": dict(res=res, want_raw=True), "pre": dict(in_slope=0.2, want_raw=False, want_act=True)}[a.mode]
        pw = cp.packed()
        run = lambda: ops.conv_mfma(x, pw, **kw)
        flops = 2.0 * n * h * w * c * c * 9
    elif a.op == "s2":
        cp = ConvParam(96, 192, 3, stride=2).cuda(); x = torch.rand(32, 256, 256, 96, device=dev) - 0.5
        pw = cp.packed()
        run = lambda: ops.conv_mfma(x, pw, stride=2, want_raw=True)
        flops = 2.0 * 32 * 128 * 128 * 96 * 192 * 9
    else:
        cp = ConvParam(192, 96, 2, transposed=True, stride=2).cuda(); x = torch.rand(32, 128, 128, 192, device=dev) - 0.5
        br = torch.rand(32, 256, 256, 96, device=dev) - 0.5
        pw = cp.packed()
        run = lambda: ops.conv_mfma(x, pw, res=br, want_raw=True)
        flops = 2.0 * 32 * 128 * 128 * 192 * 96 * 4
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    with PowerSampler(0, period=0.01) as ps:
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < a.seconds:
            for _ in range(100):
                run()
            torch.cuda.synchronize(); k += 100
        dt = time.perf_counter() - t0
    p = ps.summary(skip_s=0.3)
    ms = dt / k * 1e3
    row = {"tag": a.tag, "op": a.op, "shape": a.shape, "mode": a.mode, "ms": round(ms, 4), "w": p["socket_w_mean"], "w_max": p["socket_w_max"],
           "sclk": p["sclk_mhz_mean"], "idle_w": idle["socket_w_mean"], "j": round(p["socket_w_mean"] * ms * 1e-3, 4),
           "j_dyn": round((p["socket_w_mean"] - idle["socket_w_mean"]) * ms * 1e-3, 4), "tflops_alg": round(flops / ms / 1e9, 1),
           "cap_w": p["cap_w"], "launches": k}
    print("LEDGER " + json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="l0"); ap.add_argument("--mode", default="pre"); ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--op", default="conv", choices=["conv", "s2", "convt"])
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("VIRNET_HIP_LIB", "shipped")))
    ap.add_argument("--sweep", default=None, help="comma list of ledger builds (libvirnet_hip_led_<name>.so); 'shipped' = the product library")
    ap.add_argument("--shapes", default="l0"); ap.add_argument("--modes", default="pre,res"); ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--env", default="", help="K=V,K=V applied to every child (A/B of shipped knobs)")
    a = ap.parse_args()
    if not a.sweep:
        return one(a)
    rows = []
    extra = dict(kv.split("=", 1) for kv in a.env.split(",") if kv)
    for rep in range(a.repeat):
        for shape in a.shapes.split(","):
            for mode in a.modes.split(","):
                for name in a.sweep.split(","):
                    env = dict(os.environ, **extra)
                    if name != "shipped":
                        env["VIRNET_HIP_LIB"] = os.path.join(ROOT, "virnet_amd", "lib", f"libvirnet_hip_led_{name}.so")
                    cmd = [sys.executable, os.path.abspath(__file__), "--shape", shape, "--mode", mode, "--seconds", str(a.seconds), "--tag", name, "--op", a.op]
                    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
                    got = [json.loads(ln[7:]) for ln in out.stdout.splitlines() if ln.startswith("LEDGER ")]
                    if not got:
                        print(f"# {name} {shape} {mode}: FAILED {out.stderr[-300:]!r}", flush=True)
                        continue
                    rows.append(got[0])
                    r = got[0]
                    print(f"{r['tag']:10s} {r['shape']:3s} {r['mode']:3s}  {r['ms']:7.4f} ms  {r['w']:7.1f} W (max {r['w_max']:6.1f}, idle {r['idle_w']:5.1f})  "
                          f"sclk {r['sclk']:6.0f} MHz  J {r['j']:.4f}  J_dyn {r['j_dyn']:.4f}  {r['tflops_alg']:6.1f} TF alg", flush=True)
    print("ROWS " + json.dumps(rows), flush=True)


if __name__ == "__main__":
    main()
