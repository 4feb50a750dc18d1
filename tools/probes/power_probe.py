#!/usr/bin/env python
"""Socket power / clocks while ONE conv launch repeats (is the 96-channel conv power-capped?):
    python tools/probes/power_probe.py --shape l0 --mode pre [--seconds 4]
Samples rocm-smi in a thread while the launch loops; prints launches/s and the samples."""
import argparse, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
from bench_conv import SHAPES  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="l0"); ap.add_argument("--mode", default="pre"); ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--op", default="conv", choices=["conv", "s2", "convt"], help="s2: the 96->192 stride-2 conv at 32x256^2; convt: 192->96 transposed + bridge at 32x128^2")
    a = ap.parse_args()
    if a.op != "conv":
        if a.op == "s2":
            cp = ConvParam(96, 192, 3, stride=2).cuda(); x = torch.rand(32, 256, 256, 96, device="cuda") - 0.5
            run = lambda: ops.conv_mfma(x, cp.packed(), stride=2, want_raw=True)
        else:
            cp = ConvParam(192, 96, 2, transposed=True, stride=2).cuda(); x = torch.rand(32, 128, 128, 192, device="cuda") - 0.5
            br = torch.rand(32, 256, 256, 96, device="cuda") - 0.5
            run = lambda: ops.conv_mfma(x, cp.packed(), res=br, want_raw=True)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        samples, stop = [], threading.Event()
        def sample2():
            while not stop.is_set():
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
                samples.append(out.strip().split("\n")[-1])
                time.sleep(0.3)
        th = threading.Thread(target=sample2); th.start()
        t0 = time.time(); k = 0
        while time.time() - t0 < a.seconds:
            for _ in range(50):
                run()
            torch.cuda.synchronize(); k += 50
        dt = time.time() - t0
        stop.set(); th.join()
        print(f"{a.op}: {dt / k * 1e3:.3f} ms/launch back to back")
        for s_ in samples[2:5]:
            print("   ", s_[-60:])
        return
    n, h, w, c = SHAPES[a.shape]
    cp = ConvParam(c, c, 3).cuda()
    x = torch.rand(n, h, w, c, device="cuda") - 0.5
    res = torch.rand(n, h, w, c, device="cuda") - 0.5
    z = os.environ.get("BENCH_ZEROS", "0")       # 1: everything zero; x: zero activations, random weights; w: zero weights, random activations
    if z in ("1", "x"):
        x.zero_(); res.zero_()
    if z in ("1", "w"):
        with torch.no_grad():
            cp.weight.zero_(); cp.bias.zero_()
    kw = {"res": dict(res=res, want_raw=True), "pre": dict(in_slope=0.2, want_raw=False, want_act=True)}[a.mode]
    pw = cp.packed()
    for _ in range(5):
        ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()
    def sample():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
                samples.append(out.strip().replace("\n", " | "))
            except Exception as e:  # noqa: BLE001
                samples.append(repr(e))
            time.sleep(0.3)
    th = threading.Thread(target=sample); th.start()
    t0 = time.time(); k = 0
    while time.time() - t0 < a.seconds:
        for _ in range(50):
            ops.conv_mfma(x, pw, **kw)
        torch.cuda.synchronize(); k += 50
    dt = time.time() - t0
    stop.set(); th.join()
    flops = 2.0 * n * h * w * c * c * 9
    print(f"{a.shape} {a.mode} rows={os.environ.get('VIRNET_WX4_ROWS')} zeros={os.environ.get('BENCH_ZEROS')}: {dt / k * 1e3:.3f} ms/launch back to back, {flops * k / dt / 1e12:.1f} TFLOP/s algorithmic")
    for s in samples[2:6]:
        print("   ", s[:400])

if __name__ == "__main__":
    main()
