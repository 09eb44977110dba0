#!/usr/bin/env python3
"""Development aid: compile timing-only variants of csrc/slam2d.hip (pieces of a kernel switched off by text substitution
on a scratch copy; results are WRONG, only the stage times mean anything) into build_abl/*.so.  Run them on the GPU box
with  SLAM2D_LIB=build_abl/<name>.so python bench.py ...  (tools/ablate_run.sh)."""
import os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(REPO, "slam-2d-lidar-scan_amd/csrc/slam2d.hip")).read()
VARIANTS = {
    "base": [],
    "upd8": [("#define UPDB_UNROLL 4", "#define UPDB_UNROLL 8")],
    "upd6": [("#define UPDB_UNROLL 4", "#define UPDB_UNROLL 6")],
    "blur4w": [("__global__ __launch_bounds__(BLUR_THREADS) void k_blur_clamp(Slam2dLevel lv) {", "__global__ __launch_bounds__(BLUR_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_blur_clamp(Slam2dLevel lv) {")],
    "sc64": [("#define SCATTER_ROWS 32", "#define SCATTER_ROWS 64")],
    "sc16": [("#define SCATTER_ROWS 32", "#define SCATTER_ROWS 16")],
    "ep_nomark": [("        if (mark) {                                        // tiles of the", "        if (false) {                                       // tiles of the")],
    "ep_nohash": [("""        for (;;) {
            const int prev = atomicCAS(&hkey[h], INT_MAX, key[q]);
            if (prev == INT_MAX || prev == key[q]) break;
            h = (h + 1) & hmask;
        }
        slot[q] = h;
        atomicMin(&hown[h], q * NT + tid);""", """        h = (q * NT + tid) & hmask; hkey[h] = key[q];
        slot[q] = h;
        hown[h] = q * NT + tid;""")],
    "ep_nostore": [("                    out[pos] = y0 * lv.fpitch + x0;\n                    if (pout) {", "                    if (pout && pos < 0) {")],
    "ep_nokeys": [("                const double qx = ex + c * dx - s * dy;                             // :169", "                const double qx = ex + dx;"),
                  ("                const double qy = ey + s * dx + c * dy;                             // :170", "                const double qy = ey + dy;")],
}
def main():
    names = sys.argv[1:] or list(VARIANTS)
    os.makedirs(os.path.join(REPO, "build_abl"), exist_ok=True)
    procs = []
    for n in names:
        s = SRC
        for a, b in VARIANTS[n]:
            assert s.count(a) == 1, (n, a[:40], s.count(a))
            s = s.replace(a, b)
        f = os.path.join(tempfile.gettempdir(), f"abl_{n}.hip")
        open(f, "w").write(s)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-I", os.path.join(REPO, "include"), f, "-o", os.path.join(REPO, "build_abl", n + ".so")]
        procs.append((n, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for n, p in procs:
        out, _ = p.communicate()
        print(n, "ok" if p.returncode == 0 else "FAILED\n" + out[-2000:])
if __name__ == "__main__":
    main()
