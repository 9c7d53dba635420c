#!/usr/bin/env python3
"""Reproducer for the capture fault the library works around (csrc/vslam_hip.hip, vslam_detect_batch_dev: "nested forks").

One vslam_detect_batch_dev call of the ORIENTED mode is captured into a hipGraph after a warm-up call.  In that mode
the list stream (a side stream) forks twice to other side streams and takes the joins back: the early edge test
(enqueue_edge_flags_early: ev_list0 / ev_edge) and the spread orientation launches (enqueue_orient_batch: ev_or_fork /
ev_or_join).  The library keeps both on the list stream while a capture is on; VSLAM_CAPTURE_NESTED_FORKS=1 leaves them
in.  Each case runs in a child process (a host-side crash must not take the others down) and reports the HIP status of
every step:

    python tools/graph_try.py                 # all cases
    python tools/graph_try.py --case raw 1    # one case in this process: (capture API, nested forks 0/1)

capture API: "torch" = torch.cuda.graph (hipStreamBeginCapture in GLOBAL mode + torch's own bookkeeping),
"raw" = hipStreamBeginCapture(ThreadLocal) / hipStreamEndCapture / hipGraphInstantiate / hipGraphLaunch through ctypes on
the process's own libamdhip64.
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hip_lib():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    return C.CDLL("libamdhip64.so")


def one(api, nested, n=8, rows=240, cols=320, spread=1, early=1):
    import faulthandler

    faulthandler.enable()  # the Python-level stack of a crash (its own alternate signal stack), in front of tools/segv_bt.so's native one
    import numpy as np
    import torch

    from visualslam_amd import capi, synth

    capi.build()
    dev = "cuda:0"
    st = torch.cuda.Stream()
    rep = {"api": api, "nested_forks_in_capture": int(nested), "frames": n, "steps": []}

    def step(name, fn):
        print("GRAPH_TRY_STEP " + name, file=sys.stderr, flush=True)  # the last line before a crash names the step
        try:
            r = fn()
            rep["steps"].append({name: "ok" if r in (None, 0) else r})
            return True
        except Exception as e:  # capi raises VslamError with the library's message (HIP call + hipGetErrorString)
            rep["steps"].append({name: f"{type(e).__name__}: {str(e)[:300]}"})
            return False

    with torch.cuda.stream(st):
        ctx = capi.Context(0, st.cuda_stream)
        p = capi.default_params(rows, cols, n_octaves=3, localize=1, orient=1)
        L = capi.batch_layout(p)
        frames = torch.from_numpy(np.stack([synth.frame_np(rows, cols, f, 20, "noise") for f in range(n)])).to(dev)

        def outs():
            return dict(response=torch.zeros((n, rows, cols), dtype=torch.float32, device=dev),
                        harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                        pyramid=torch.zeros((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                        dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                        oriented_points=torch.zeros((n, p.oriented_cap, 6), dtype=torch.int32, device=dev), oriented_counts=torch.zeros(n, dtype=torch.int32, device=dev))

        a, b = outs(), outs()
        ctx.detect_batch(p, frames, **a)  # warm-up: workspace, tables, side streams
        torch.cuda.synchronize()
        ok = True
        if api == "torch":
            g = torch.cuda.CUDAGraph()

            def cap():
                with torch.cuda.graph(g, stream=st):
                    ctx.detect_batch(p, frames, **b)

            ok = step("capture (torch.cuda.graph)", cap)
            ok = ok and step("replay", lambda: (g.replay(), torch.cuda.synchronize())[0])
        else:
            hip = hip_lib()
            hip.hipGetErrorString.restype = C.c_char_p
            hip.hipGetErrorName.restype = C.c_char_p

            def call(name, fn):
                print("GRAPH_TRY_STEP calling " + name, file=sys.stderr, flush=True)  # the last "calling" line before a crash names the API
                rc = fn()
                print("GRAPH_TRY_STEP returned " + name, file=sys.stderr, flush=True)
                rep["steps"].append({name: "hipSuccess" if rc == 0 else f"{rc} {hip.hipGetErrorName(rc).decode()}: {hip.hipGetErrorString(rc).decode()}"})
                return rc == 0

            sh = C.c_void_p(st.cuda_stream)
            graph, gexec = C.c_void_p(), C.c_void_p()
            ok = call("hipStreamBeginCapture(ThreadLocal)", lambda: hip.hipStreamBeginCapture(sh, 1))
            ok = ok and step("vslam_detect_batch_dev under capture", lambda: ctx.detect_batch(p, frames, **b))
            print("GRAPH_TRY_STEP returned vslam_detect_batch_dev", file=sys.stderr, flush=True)
            ok = call("hipStreamEndCapture", lambda: hip.hipStreamEndCapture(sh, C.byref(graph))) and ok
            if ok:
                nn = C.c_size_t()
                call("hipGraphGetNodes", lambda: hip.hipGraphGetNodes(graph, None, C.byref(nn)))
                rep["graph_nodes"] = nn.value
                ok = call("hipGraphInstantiate", lambda: hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0))
            if ok:
                ok = call("hipGraphLaunch", lambda: hip.hipGraphLaunch(gexec, sh))
                ok = call("hipStreamSynchronize", lambda: hip.hipStreamSynchronize(sh)) and ok
        if ok:
            torch.cuda.synchronize()
            same = bool(torch.equal(a["oriented_counts"], b["oriented_counts"]) and torch.equal(a["dog_counts"], b["dog_counts"]))
            rep["replay_equals_eager"] = same
            rep["oriented_points"] = int(b["oriented_counts"].sum())
        rep["ok"] = bool(ok)
    print("GRAPH_TRY " + json.dumps(rep), flush=True)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--case":
        one(sys.argv[2], int(sys.argv[3]))
        return
    for api in ("raw", "torch"):
        for nested in ((0, 2, 3, 1) if api == "raw" else (0, 1)):  # 2: only the early edge test's fork, 3: only the spread launches', 1: both
            # the switch exists in the diagnostics build only (lib/libvslam_diag.so, -DVSLAM_DIAGNOSTICS): the shipped library cannot
            # be talked into the crash
            env = dict(os.environ, VSLAM_CAPTURE_NESTED_FORKS=str(nested), VSLAM_LIBRARY=os.path.join(ROOT, "visualslam_amd", "lib", "libvslam_diag.so"))
            bt = os.path.join(ROOT, "tools", "segv_bt.so")  # gcc -shared -fPIC -O1 -g tools/segv_bt.c -o tools/segv_bt.so
            if os.path.exists(bt):
                env["LD_PRELOAD"] = (env.get("LD_PRELOAD", "") + " " + bt).strip()
            r = subprocess.run(["timeout", "-k", "10", "120", sys.executable, os.path.abspath(__file__), "--case", api, str(nested)], capture_output=True, text=True, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("GRAPH_TRY ")]
            print(json.dumps({"case": [api, nested], "exit_code": r.returncode, "report": json.loads(lines[-1][10:]) if lines else None,
                              "steps_seen": [l for l in r.stderr.splitlines() if l.startswith("GRAPH_TRY_STEP")][-4:],
                              "stderr_tail": r.stderr[-6000:] if (r.returncode != 0 or not lines) else ""}), flush=True)


if __name__ == "__main__":
    main()
