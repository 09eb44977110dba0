#!/usr/bin/env python3
"""Development aid: compile timing-only variants of csrc/slam2d.hip (pieces of a kernel switched off by text substitution
on a scratch copy; results are WRONG, only the stage times mean anything) into build_abl/*.so.  Run them on the GPU box
with  SLAM2D_LIB=build_abl/<name>.so python bench.py ...  (tools/ablate_run.sh)."""
import os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(REPO, "slam-2d-lidar-scan_amd/csrc/slam2d.hip")).read()
VARIANTS = {
    "base": [],
    "ep_nomark": [("        if (mark) {                                        // tiles of the", "        if (false) {                                       // tiles of the")],
    "ep_nodiv": [("const int cx = (int)((qx - fr.xlo) / lv.step);", "const int cx = (int)((qx - fr.xlo) * (1.0 / lv.step));"),
                 ("const int cy = (int)((qy - fr.ylo) / lv.step);", "const int cy = (int)((qy - fr.ylo) * (1.0 / lv.step));")],
    "ep_nohash": [("""        for (;;) {
            const int prev = atomicCAS(&hkey[h], INT_MAX, key[q]);
            if (prev == INT_MAX || prev == key[q]) break;
            h = (h + 1) & hmask;
        }
        slot[q] = h;
        atomicMin(&hown[h], q * 256 + tid);""", """        h = (q * 256 + tid) & hmask; hkey[h] = key[q];
        slot[q] = h;
        hown[h] = q * 256 + tid;""")],
    "ep_nomin": [("        atomicMin(&hown[h], q * 256 + tid);", "        hown[h] = q * 256 + tid;")],
    "bg1": [("#define BOUND_GROUP 4 ", "#define BOUND_GROUP 1 ")],
    "bg2": [("#define BOUND_GROUP 4 ", "#define BOUND_GROUP 2 ")],
    "bg8": [("#define BOUND_GROUP 4 ", "#define BOUND_GROUP 8 ")],
    "ep_nostore": [("                    out[pos] = key[q];\n                    if (pout) {", "                    if (pout && pos < 0) {")],
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
