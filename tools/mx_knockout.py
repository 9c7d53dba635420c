"""Knock-out builds of the matrix path's octave-0 kernel (round 5): which part of k_pyr_octave_mx<MxCfgOct0> its time
belongs to.  Each variant is a copy of visualslam_amd/csrc under /tmp with ONE part of the kernel replaced by a register
sink (results are wrong by construction - these libraries are for timing only and never leave visualslam_amd/lib/ab/),
compiled as the octave-0 translation unit and linked with the product's other objects.

    python tools/mx_knockout.py build [variant ...]            # here (no GPU): visualslam_amd/lib/ab/ko_<variant>.so
    python tools/mx_knockout.py run [variant ...] [--steps 5]  # on the GPU box: tools/mx_alone.py --octaves 1 under each library
"""
import json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visualslam_amd", "csrc")
AB = os.path.join(ROOT, "visualslam_amd", "lib", "ab")
OBJ = os.path.join(ROOT, "visualslam_amd", "lib", "obj")

SINK4 = 'asm volatile("" ::"v"({0}.x), "v"({0}.y), "v"({0}.z), "v"({0}.w));'

STORE = """                *reinterpret_cast<uint4*>(gp + ln.off + i * ln.pitch8) = gv;
                if (L > 0) *reinterpret_cast<uint4*>(dp + ln.off + i * ln.pitch8) = dv;"""
LDSW = """        *reinterpret_cast<uint4*>(ln.wb + 8 * ob) = make_uint4(g[0], g[1], g[2], g[3]);
        if (CFG::DBUF && L > 0) *reinterpret_cast<uint4*>(ln.wb + CFG::OBUF + 8 * ob) = make_uint4(dd[ob][0], dd[ob][1], dd[ob][2], dd[ob][3]);"""
FLUSH_HEAD = "    if (CFG::DBUF) {\n        uint8_t* dp = ln.out + (size_t)(VSLAM_NUM_LEVELS + L - 1) * ln.P;"
STAGE = "        mx_stage_tile_up2<CFG::TW, CFG::TH, CFG::R, CFG::RWP, CFG::NT>(base + fz * bframe, sstep, rows / 2, cols / 2, tile_x0, tile_y0, smem, 0x80808080u);"
EPI = """            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = ((uint32_t)chi[4 * k + j] << 8) + (uint32_t)clo[4 * k + j];
            const uint32_t e = __builtin_amdgcn_perm(w[2], w[0], 0x0c060c02);  // (G0, G2) in 16-bit lanes
            const uint32_t o = __builtin_amdgcn_perm(w[3], w[1], 0x0c060c02);  // (G1, G3)
            g[k] = __builtin_amdgcn_perm(o, e, 0x06020400);
            if (L > 0)  // D_{L-1} = saturate_u8(G_L - G_{L-1}), GaussPyramid.cpp:197
                dd[ob][k] = __builtin_amdgcn_perm(mx_pk_sub_sat_u16(o, po[ob][k]), mx_pk_sub_sat_u16(e, pe[ob][k]), 0x06020400);
            pe[ob][k] = e;
            po[ob][k] = o;"""
SPLIT = """            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)c1[4 * d + 1], (uint32_t)c1[4 * d + 0], 0x05010400);  // (lo0, lo1, hi0, hi1)
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)c1[4 * d + 3], (uint32_t)c1[4 * d + 2], 0x05010400);
            lo[slot][d] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100) ^ 0x80808080u);
            hi[slot][d] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302);"""

VARIANTS = {
    "base": [],
    # no HBM stores of the planes (the LDS round trip stays)
    "nostore": [(STORE, "                " + SINK4.format("gv") + "\n                if (L > 0) { " + SINK4.format("dv") + " }")],
    # G planes only (half the bytes)
    "halfstore": [(STORE, "                *reinterpret_cast<uint4*>(gp + ln.off + i * ln.pitch8) = gv;\n                if (L > 0) { " + SINK4.format("dv") + " }")],
    # no tile staging (the upsample and the LDS fill): the levels run on whatever the LDS holds
    "nostage": [(STAGE, "        ;")],
    # epilogue (shift-add, byte picks, DoG) replaced by one op per output dword
    "noepi": [(EPI, """            g[k] = (uint32_t)chi[4 * k] ^ (uint32_t)clo[4 * k + 1] ^ (uint32_t)chi[4 * k + 2] ^ (uint32_t)clo[4 * k + 3] ^ (uint32_t)chi[4 * k + 1] ^ (uint32_t)clo[4 * k] ^ (uint32_t)chi[4 * k + 3] ^ (uint32_t)clo[4 * k + 2];
            if (L > 0) dd[ob][k] = g[k] ^ pe[ob][k];
            pe[ob][k] = g[k];
            po[ob][k] = g[k];""")],
    # hand-off split replaced by two moves per four values
    "nosplit": [(SPLIT, """            lo[slot][d] = c1[4 * d] ^ c1[4 * d + 2];
            hi[slot][d] = c1[4 * d + 1] ^ c1[4 * d + 3];""")],
}
VARIANTS["nostore_nostage"] = VARIANTS["nostore"] + VARIANTS["nostage"]
VARIANTS["nostore_noepi_nosplit"] = VARIANTS["nostore"] + VARIANTS["noepi"] + VARIANTS["nosplit"]
VARIANTS["all_out"] = VARIANTS["nostore"] + VARIANTS["nostage"] + VARIANTS["noepi"] + VARIANTS["nosplit"]
# staggered first round: the two workgroups of a CU start together and - every tile taking the same time - stay in step, both
# staging, then both computing; delaying one of them by about half a tile once, in the first round only, would keep them apart
ENTRY = "    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];\n    // XCD-aware tile order, as k_pyr_octave"
def stagger(cond, sleeps):
    return [(ENTRY, """    {
        const unsigned int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), j = lin >> 3;
        if (%s)
            for (int i = 0; i < %d; ++i) __builtin_amdgcn_s_sleep(127);
    }
""" % (cond, sleeps) + ENTRY)]
VARIANTS["stagger_hi3"] = stagger("j >= 32 && j < 64", 3)
VARIANTS["stagger_hi2"] = stagger("j >= 32 && j < 64", 2)
VARIANTS["stagger_hi4"] = stagger("j >= 32 && j < 64", 4)
VARIANTS["stagger_odd3"] = stagger("j < 64 && (j & 1)", 3)
VARIANTS["stagger_all_odd1"] = stagger("(j & 1)", 1)
DEFINES = {}  # variant -> extra -D flags (round 5 tried MX_PREFETCH_TILES, MX_TAPS_EARLY, MX_FLUSH_WIDE; none paid, the macros are gone)


def build(only):
    os.makedirs(AB, exist_ok=True)
    for name, patches in VARIANTS.items():
        if only and name not in only:
            continue
        d = os.path.join("/tmp", "mx_ko", name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(os.path.join(d, "visualslam_amd"))
        shutil.copytree(CSRC, os.path.join(d, "visualslam_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        h = os.path.join(d, "visualslam_amd", "csrc", "kernels_pyramid_mx.hip.h")
        s = open(h).read()
        for old, new in patches:
            assert s.count(old) == 1, (name, old[:60], s.count(old))
            s = s.replace(old, new)
        open(h, "w").write(s)
        o = os.path.join(d, "mx0.o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm",
               "-amdgpu-sched-strategy=max-ilp"] + DEFINES.get(name, []) + ["-c", "-o", o, os.path.join(d, "visualslam_amd", "csrc", "vslam_mx0.hip")]
        subprocess.check_call(cmd)
        out = os.path.join(AB, "ko_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, os.path.join(OBJ, "vslam_hip.o"),
                               os.path.join(OBJ, "vslam_params.o"), os.path.join(OBJ, "vslam_mx.o"), o])
        print("built", out, flush=True)


def run(steps, only):
    res = {}
    for name in VARIANTS:
        if only and name not in only:
            continue
        lib = os.path.join(AB, "ko_%s.so" % name)
        env = dict(os.environ, VSLAM_MX="1", VSLAM_LIBRARY=lib)
        for rep in range(2):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mx_alone.py"), "--octaves", "1", "--steps", str(steps)], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(name, "failed", out.stderr[-400:], flush=True)
                break
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res.setdefault(name, []).append(round(d["octave_kernel_ms_per_step"], 3))
        print(name, res.get(name), flush=True)
    print(json.dumps({"what": "k_pyr_octave_mx<MxCfgOct0, false, true> alone (256 x 1080p, pyramid only), ms per launch, knock-out builds", "ms": res}))


if __name__ == "__main__":
    only = [a for a in sys.argv[2:] if not a.startswith("--") and not a.isdigit()]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 5
    if sys.argv[1] == "build":
        build(only)
    else:
        run(steps, only)
