#!/usr/bin/env python
"""What may another host thread do while this one captures a hipGraph (capture_error_mode="thread_local")?  (round 6: the automatic capture of
net(x) aborted the process once in ~3 runs of a two-thread test.)  Thread A captures the denoiser's forward N times; thread B meanwhile loops ONE
kind of activity.  Prints, per activity, how many of A's captures raised.

    python tools/probes/capture_concurrency.py [--n 30]"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import SYN_CFG  # noqa: E402
from virnet_amd import engine, graph  # noqa: E402
from virnet_amd.networks import VIRAttResUNet  # noqa: E402
from virnet_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if not a.only:
        # every activity in a process of its own: two of them end in an exception thrown from a destructor, i.e. abort
        import subprocess
        for name in ("nothing", "stream.synchronize", "tensor.item", "event record", "eager forward", "alloc + empty_cache", "pin_memory", "replay of its own",
                     "captures of its own", "capture, drop"):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--n", str(a.n), "--only", name], capture_output=True, text=True)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("B:")]
            what = [ln for ln in p.stderr.splitlines() if "what()" in ln]
            print((lines[0] if lines else f"B: {name:28s} -> no result") + ("" if p.returncode == 0 else f"   [process ended with {p.returncode}" + (": " + what[0].strip() if what else "") + "]"), flush=True)
        return
    net = VIRAttResUNet(im_chn=3, sigma_chn=1, **SYN_CFG)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}))
    net = net.cuda().eval()
    x = synth_images(1, 3, 64, 64).cuda()
    xb = synth_images(1, 3, 48, 80).cuda()
    with torch.no_grad(), graph.no_autograph():
        for _ in range(3):
            net(x); net(xb)
    torch.cuda.synchronize()
    stop = threading.Event()

    def act_none():
        time.sleep(0.001)

    def act_sync():
        torch.cuda.current_stream().synchronize()

    def act_item():
        torch.ones(1, device="cuda").item()

    def act_eager():
        with torch.no_grad(), graph.no_autograph():
            net(xb)

    def act_alloc():
        t = torch.empty(1 << 22, device="cuda"); del t
        torch.cuda.empty_cache()

    def act_pin():
        t = torch.zeros(1 << 10).pin_memory(); del t

    def act_event():
        e = torch.cuda.Event(); e.record(); e.synchronize()

    gb = graph.GraphedForward(lambda t: engine.denoise_forward(net, t), warmup=1, check="sync")
    with torch.no_grad():
        gb(xb)
    torch.cuda.synchronize()

    def act_replay():
        with torch.no_grad():
            gb(xb)

    def act_capture():
        g2 = graph.GraphedForward(lambda t: engine.denoise_forward(net, t), warmup=1, check="sync")
        with torch.no_grad():
            g2(xb)
        torch.cuda.synchronize()
        g2.reset()

    def act_gc():
        import gc
        g2 = graph.GraphedForward(lambda t: engine.denoise_forward(net, t), warmup=1, check="sync")
        with torch.no_grad():
            g2(xb)
        del g2                      # dropped WITHOUT reset(): its graphs go to the graveyard
        gc.collect()

    acts = {"nothing": act_none, "capture, drop without reset, gc.collect": act_gc, "replay of its own graph": act_replay, "captures of its own (serialised by the lock)": act_capture, "stream.synchronize": act_sync, "tensor.item": act_item, "event record+sync": act_event, "pin_memory allocation": act_pin,
            "eager forward (own stream)": act_eager, "alloc + empty_cache": act_alloc}
    for name, fn in acts.items():
        if a.only and a.only not in name:
            continue
        stop.clear()
        berr = []

        def bwork():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                while not stop.is_set():
                    try:
                        fn()
                    except Exception as e:          # noqa: BLE001
                        berr.append(repr(e)[:120])
                        break

        tb = threading.Thread(target=bwork)
        tb.start()
        bad, first = 0, None
        for i in range(a.n):
            try:
                gf = graph.GraphedForward(lambda t: engine.denoise_forward(net, t), warmup=1, check="sync")
                with torch.no_grad():
                    gf(x)
                torch.cuda.synchronize()
                gf.reset()
            except Exception as e:                  # noqa: BLE001
                bad += 1
                first = first or repr(e)[:160]
        stop.set()
        tb.join()
        print(f"B: {name:28s} -> {bad} of {a.n} captures raised" + (f"  first: {first}" if first else "") + (f"  B raised: {berr[0]}" if berr else ""), flush=True)


if __name__ == "__main__":
    main()
