"""Driver of tools/mfma_probe.hip (not part of the product, not on the default path):
builds the probe, runs one pyramid level as two chained i8 MFMA products on the GPU, checks the result bit
for bit against oracle.gaussian_blur_u8 and prints what it cost next to the packed-dot figures of the product
kernels.

    python tools/mfma_probe.py [--planes 32] [--reps 10] > profiles/r03_mfma_probe.txt
"""
import argparse, json, os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from visualslam_amd import capi, synth


def trimmed(taps):
    nz = taps.nonzero()[0]
    return taps[nz[0]: nz[-1] + 1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    exe = os.path.join(ROOT, "tools", "mfma_probe")
    src = exe + ".hip"
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        # accumulators in VGPRs (gfx950 has one unified file): without the flag every value goes through v_accvgpr_read before the VALU can touch it
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-mllvm", "-amdgpu-mfma-vgpr-form", "-o", exe, src], check=True)
    capi.build()
    oracle.build()
    # (octave, level) -> plane size of a 1080p frame's pyramid, the level's sigma
    cases = [("octave 0 level 5 (widest of the 3840x2160 octave)", 2160, 3840, 0, 5), ("octave 0 level 0", 2160, 3840, 0, 0),
             ("octave 1 level 5 (widest of the 1920x1080 octave)", 1080, 1920, 1, 5), ("octave 1 level 2", 1080, 1920, 1, 2)]
    results = []
    for name, rows, cols, o, l in cases:
        sigma = capi.sigma_at(1.6, o, l)
        ks = capi.gauss_ksize_u8(sigma)
        taps = trimmed(capi.gauss_taps_q8(ks, sigma))
        img = synth.frame_np(rows, cols, 0, 11, "noise" if l == 5 else "checker")
        with tempfile.TemporaryDirectory() as td:
            fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
            with open(fin, "wb") as f:
                f.write(np.array([rows, cols, len(taps)], np.int32).tobytes() + taps.astype(np.uint8).tobytes() + img.tobytes())
            planes = a.planes if rows > 1080 else 4 * a.planes
            r = subprocess.run([exe, fin, fout, str(planes), str(a.reps)], capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit("probe failed: " + r.stdout + r.stderr)
            res = json.loads(r.stdout.strip().splitlines()[-1])
            got = np.fromfile(fout, np.uint8).reshape(rows, cols)
        want = oracle.gaussian_blur_u8(img, ks, sigma)
        res["case"] = name
        res["sigma"] = sigma
        res["ksize_opencv"] = ks
        res["bit_exact_vs_oracle"] = bool((got == want).all())
        res["mismatching_pixels"] = int((got != want).sum())
        results.append(res)
    print("MFMA feasibility probe: one Gaussian level as two chained v_mfma_i32_32x32x32_i8 (tools/mfma_probe.hip)")
    print("NOT wired into the product: BASELINE.json's north star rules MFMA out; measured for the rule's owner.\n")
    for r in results:
        print(json.dumps(r))
    print()
    # the product kernels' figures for the same work (measured issue costs, 1024 SIMDs):
    # dot4 0.52 ns and dot2 0.93 ns per MAC per SIMD-lane-group -> per pixel and level n/4 dot4 + n/2 dot2 wave-instructions / 64 lanes
    print("per level, MFMA probe vs the packed-dot floor of the product kernel (2.0 ns per wave64 dot instruction per SIMD, 1024 SIMDs):")
    for r in results:
        n = r["taps"]
        px = r["rows"] * r["cols"] * r["planes"]
        dot_floor_ms = px / 64 * (n / 4 + n / 2) * 2.0e-9 / 1024 * 1e3
        print(f"  {r['case']}: {n} taps, {r['planes']} planes: MFMA {r['ms_per_launch']:.3f} ms ({r['gpixel_per_s']:.1f} Gpixel/s, "
              f"{r['hbm_GBps_in_plus_out']:.0f} GB/s in+out), dot4+dot2 floor {dot_floor_ms:.3f} ms -> {dot_floor_ms / r['ms_per_launch']:.2f}x; "
              f"bit-exact: {r['bit_exact_vs_oracle']}")
        m = r["ms_marginal_per_level"]
        print(f"      six levels on one staged tile: {r['ms_six_levels']:.3f} ms ({r['six_levels_GBps_in_plus_6out']:.0f} GB/s for 1 plane in + 6 out) -> marginal cost of a level "
              f"{m:.3f} ms = {dot_floor_ms / m:.2f}x the dot floor's speed; issued MFMA rate in that margin: "
              f"{r['issued_mfma'] / r['planes'] * 0 + (r['mfma_per_32x32_block_interior'] * r['rows'] * r['cols'] * r['planes'] / 1024) * 32768 / (m * 1e-3) / 2.4e9 / 256:.0f} MAC/clk/CU of 4096")


if __name__ == "__main__":
    main()
