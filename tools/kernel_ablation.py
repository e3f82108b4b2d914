#!/usr/bin/env python3
"""Where the time goes INSIDE the Winograd kernels (3x3 forward / backward-data, 3x3 backward-weights, the folded decoders' F(2x2,4x4)):
timing builds with one ingredient of the main loop removed.

The script copies csrc/ to a scratch directory, patches `conv_wino.hip` (forward / backward-data) and `conv_wgrad_wino.hip`
(backward-weights) so that a compile-time bit mask ABL switches parts of the loop off — the results of such a build are WRONG, its
duration is what is measured — and links one library per mask into rpg_ramnet_amd/abl/ (git-ignored, travels with gpurun).
`RAMNET_HIP_LIB=<lib> python tools/bench_layers.py --only gru` then times the six ConvGRU launches with it.

  bit   1  no staging (global loads of the next raw patch / strip, LDS stores, batch walk)
        2  no LDS reads of the raw data (operands stay what they were; made opaque to the compiler so that nothing is hoisted)
        4  no transform arithmetic
        8  no barrier
       16  no sched_barrier (the compiler orders the loop)
       32  (forward) no weight loads
       64  (forward) no epilogue (exchange + output transform + stores)
      128  no main loop at all: the launch's fixed cost
      256  (backward-weights) no atomic adds of the partial sums

Usage (container, then GPU box):
  python tools/kernel_ablation.py build [kernel ...]     # all masks of `MASKS` (optionally of the named kernels only, e.g. conv_wino24)
  gpurun -- 'python tools/kernel_ablation.py run [kernel ...]'        # -> gpurun_out/ablation.txt
Results of round 3: profiles/r03_k_ablation.txt, discussed in profiles/r03_h_tuning_notes.md."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rpg_ramnet_amd")
OUT = os.path.join(PKG, "abl")
SCRATCH = os.environ.get("RAMNET_ABL_SCRATCH", "/tmp/ramnet_abl")
MASKS = {"conv_wino6": [0, 1, 2, 4, 8, 32, 6, 7, 39, 47, 64, 128, 192],        # F(2x4,3x3): the six ConvGRU launches at B = 8 run it
         "conv_wgrad_wino": [0, 1, 2, 4, 8, 16, 3, 7, 15, 128, 256, 384],
         "conv_wino": [0, 1, 2, 4, 8, 16, 32, 3, 7, 39, 47, 64, 128, 192],
         "conv_wino24": [0, 1, 2, 4, 8, 32, 39, 47, 64, 128, 192]}        # the folded decoders: timed with `bench_layers.py --only dec`
ONLY = {"conv_wgrad_wino": "gru", "conv_wino": "gru", "conv_wino24": "dec", "conv_wino6": "gru"}


def _sub(s, old, new, count=1):
    if old not in s:
        raise RuntimeError("ablation patch does not apply (kernel source changed): %r" % old[:60])
    return s.replace(old, new) if count == 0 else s.replace(old, new, count)


def patch_wgrad(s):
    s = _sub(s, '#include "common.hpp"\n', '#include "common.hpp"\n#ifndef ABL\n#define ABL 0\n#endif\n#define OPQ(QQ) asm volatile("" : "+v"(QQ))\n')
    s = _sub(s, "for (int c = 0; c < 4; ++c) da[c] = xc[xa_off + st * G::SX + c * 32], db[c] = xc[xb_off + st * G::SX + c * 32];",
             "for (int c = 0; c < 4; ++c) { if (ABL & 2) { OPQ(da[c]); OPQ(db[c]); } else da[c] = xc[xa_off + st * G::SX + c * 32], db[c] = xc[xb_off + st * G::SX + c * 32]; }")
    s = _sub(s, "                g0[f][c] = yc[y_off + (yr0 + st * G::SY + c) * GW_CO + f * 32], g1[f][c] = yc[y_off + (YW + st * G::SY + c) * GW_CO + f * 32];",
             "                { if (ABL & 2) { OPQ(g0[f][c]); OPQ(g1[f][c]); } else g0[f][c] = yc[y_off + (yr0 + st * G::SY + c) * GW_CO + f * 32], g1[f][c] = yc[y_off + (YW + st * G::SY + c) * GW_CO + f * 32]; }")
    s = _sub(s, "    auto finish_x = [&]() {\n", "    auto finish_x = [&]() {\n        if (ABL & 4) { OPQ(an[0]); OPQ(an[1]); OPQ(an[2]); OPQ(an[3]); return; }\n")
    s = _sub(s, "    auto finish_y = [&](int f) {\n", "    auto finish_y = [&](int f) {\n        if (ABL & 4) { OPQ(bn[f][0]); OPQ(bn[f][1]); OPQ(bn[f][2]); OPQ(bn[f][3]); return; }\n")
    s = _sub(s, "            auto stage = [&](int k) {\n", "            auto stage = [&](int k) {\n                if (ABL & 1) return;\n")
    s = _sub(s, "if (st == 2) __syncthreads();", "if (st == 2 && !(ABL & 8)) __syncthreads();")
    s = _sub(s, "        for (; batch <= last; batch += step, cur ^= 1) {", "        for (; batch <= ((ABL & 128) ? -1 : last); batch += step, cur ^= 1) {")
    # (256: no join of the partial sums — the slab read-modify-write of round 4 and the atomic form)
    s = _sub(s, "                if (c < Cin && n < p.Cout) col[(size_t)c * p.Cout] = old[g & 1][r] +",
             "                if (c < Cin && n < p.Cout && (!(ABL & 256) || acc[pl][f][r] == 123.456f)) col[(size_t)c * p.Cout] = old[g & 1][r] +")
    s = _sub(s, "                    if (c < Cin && n < p.Cout) atomicAdd(col + (size_t)c * p.Cout,",
             "                    if (c < Cin && n < p.Cout && (!(ABL & 256) || acc[pl][f][r] == 123.456f)) atomicAdd(col + (size_t)c * p.Cout,")
    return _sub(s, "__builtin_amdgcn_sched_barrier(0);", "if (!(ABL & 16)) __builtin_amdgcn_sched_barrier(0);", 0)


def patch_wino(s):
    s = _sub(s, '#include "conv_epilogue.hpp"\n',
             '#include "conv_epilogue.hpp"\n#ifndef ABL\n#define ABL 0\n#endif\n#define OPQ4(QQ) asm volatile("" : "+v"((QQ).x), "+v"((QQ).y), "+v"((QQ).z), "+v"((QQ).w))\n')
    s = _sub(s, """                if (!(k & 1)) ta = ld4(pnext + pra + (k >> 1) * 4), tb = ld4(pnext + prb + (k >> 1) * 4);
                else tn[k >> 1] = te(ta, tb);
            } else if (k < 10)""", """                if (!(k & 1)) { if (ABL & 2) { OPQ4(ta); OPQ4(tb); } else ta = ld4(pnext + pra + (k >> 1) * 4), tb = ld4(pnext + prb + (k >> 1) * 4); }
                else { if (ABL & 4) { OPQ4(tn[k >> 1]); } else tn[k >> 1] = te(ta, tb); }
            } else if (ABL & 1) {
            } else if (k < 10)""")
    s = _sub(s, "            const float4 v = pl == 0 ? f4sub(tc[0], tc[2])", "            const float4 v = (ABL & 4) ? tc[pl] : pl == 0 ? f4sub(tc[0], tc[2])")
    s = _sub(s, """            breg[pl][0] = ld4(wnext + (pl * 2) * 256);
            if (NF == 2) breg[pl][NF - 1] = ld4(wnext + (pl * 2 + 1) * 256);
        }
        __syncthreads(); """, """            if (ABL & 32) { OPQ4(breg[pl][0]); OPQ4(breg[pl][NF - 1]); } else {
            breg[pl][0] = ld4(wnext + (pl * 2) * 256);
            if (NF == 2) breg[pl][NF - 1] = ld4(wnext + (pl * 2 + 1) * 256); }
        }
        if (!(ABL & 8)) __syncthreads(); """)
    s = _sub(s, "    // ---- exchange: column transform of the wave's row",
             "    if (ABL & 64) { float t = 0.f; for (int i = 0; i < 4; ++i) for (int f = 0; f < NF; ++f) for (int r = 0; r < 16; ++r) t += acc[i][f][r];"
             " if (t == 123.456f) p.out[0] = t; return; }\n    // ---- exchange: column transform of the wave's row")
    s = _sub(s, "    for (int chunk = 0; chunk < nch; chunk += 2) {", "    for (int chunk = 0; chunk < ((ABL & 128) ? 0 : nch); chunk += 2) {")
    return _sub(s, "__builtin_amdgcn_sched_barrier(0);", "if (!(ABL & 16)) __builtin_amdgcn_sched_barrier(0);", 0)


def patch_wino6(s):
    """F(2x4,3x3): 1 no patch staging, 2 no LDS reads, 4 no transform arithmetic (row combination + column transform), 8 no barrier,
    32 no weight loads, 64 no epilogue, 128 no main loop."""
    s = _sub(s, '#include "conv_wino_common.hpp"\n',
             '#include "conv_wino_common.hpp"\n#ifndef ABL\n#define ABL 0\n#endif\n#define OPQ4(QQ) asm volatile("" : "+v"((QQ).x), "+v"((QQ).y), "+v"((QQ).z), "+v"((QQ).w))\n'
             '#define OPQ1(QQ) asm volatile("" : "+v"(QQ))\n')
    s = _sub(s, """                if (!(s & 1)) qa[s >> 1] = ld4(pnext + pra + (s >> 1) * 4);
                else qb[s >> 1] = ld4(pnext + prb + (s >> 1) * 4);""",
             """                if (ABL & 2) { if (!(s & 1)) OPQ4(qa[s >> 1]); else OPQ4(qb[s >> 1]); }
                else if (!(s & 1)) qa[s >> 1] = ld4(pnext + pra + (s >> 1) * 4);
                else qb[s >> 1] = ld4(pnext + prb + (s >> 1) * 4);""")
    s = _sub(s, "                tn[j][c] = fmaf(sb, y, x);\n", "                if (ABL & 4) tn[j][c] = x; else tn[j][c] = fmaf(sb, y, x);\n")
    s = _sub(s, "            if (g + LD < 6) colop(g + LD, (g + LD + 2 * PAR) & 3, i, tc);\n            else colop(g + LD - 6, (g + LD - 6 + 2 * (PAR ^ 1)) & 3, i, tn);",
             "            if (ABL & 4) { if (i >= 4) { OPQ1(vb[(g + LD + 2 * PAR) & 3][i - 4]); } }\n"
             "            else if (g + LD < 6) colop(g + LD, (g + LD + 2 * PAR) & 3, i, tc);\n            else colop(g + LD - 6, (g + LD - 6 + 2 * (PAR ^ 1)) & 3, i, tn);")
    s = _sub(s, "            if (s >= 28 && s < 31) pr.store_slot(pfree, q.src, c2, s - 28);\n            if (s == 33 || s == 35 || s == 37) pr.load_slot(q.src, c3, (s - 33) >> 1, clast);",
             "            if (!(ABL & 1)) { if (s >= 28 && s < 31) pr.store_slot(pfree, q.src, c2, s - 28);\n            if (s == 33 || s == 35 || s == 37) pr.load_slot(q.src, c3, (s - 33) >> 1, clast); }")
    s = _sub(s, "            if ((s & 15) == 15) breg[g - 1][0] = wload(cw, (g - 1) * 2), breg[g][0] = wload(cw, g * 2);",
             "            if ((s & 15) == 15) { if (ABL & 32) { OPQ4(breg[g - 1][0]); OPQ4(breg[g][0]); } else breg[g - 1][0] = wload(cw, (g - 1) * 2), breg[g][0] = wload(cw, g * 2); }")
    s = _sub(s, "        __syncthreads();                           // patch(i+2) visible; patch(i+1) free", "        if (!(ABL & 8)) __syncthreads();")
    s = _sub(s, "    } while (chunk < nch);", "    } while (chunk < ((ABL & 128) ? 0 : nch));")
    s = _sub(s, "    // ---- exchange: column transform of the wave's row (M A4: 4 of 6 columns), all waves -> LDS.",
             "    if (ABL & 64) { float t = 0.f; for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) t += acc[i][0][r]; if (t == 123.456f) p.out[0] = t; return; }\n"
             "    // ---- exchange: column transform of the wave's row (M A4: 4 of 6 columns), all waves -> LDS.")
    return s


def patch_wino24(s):
    """1 no window loads, 2 no LDS reads of the transformed input, 4 no transform (arithmetic + LDS writes), 8 no barrier, 32 no weight
    loads, 64 no epilogue, 128 no main loop."""
    s = _sub(s, '#include "conv_epilogue.hpp"\n',
             '#include "conv_epilogue.hpp"\n#ifndef ABL\n#define ABL 0\n#endif\n#define W24_ABLATE ((ABL & 1) | ((ABL & 32) ? 2 : 0))\n'
             '#define OPQ4(QQ) asm volatile("" : "+v"((QQ).x), "+v"((QQ).y), "+v"((QQ).z), "+v"((QQ).w))\n')
    s = _sub(s, "    auto tr_col = [&](int c) {\n", "    auto tr_col = [&](int c) {\n        if (ABL & 4) return;\n")
    s = _sub(s, "    auto tr_row = [&](float *vbuf, int i) {\n", "    auto tr_row = [&](float *vbuf, int i) {\n        if (ABL & 4) return;\n")
    s = _sub(s, "                if (pos + 2 < 25) aq[(pp + 1) & 1][0] = ldv(vb + (pos + 2) * W24_PS + aoff);\n"
                "                if (pos + 3 < 25) aq[(pp + 1) & 1][1] = ldv(vb + (pos + 3) * W24_PS + aoff);",
             "                if (ABL & 2) { OPQ4(aq[(pp + 1) & 1][0]); OPQ4(aq[(pp + 1) & 1][1]); } else {\n"
             "                if (pos + 2 < 25) aq[(pp + 1) & 1][0] = ldv(vb + (pos + 2) * W24_PS + aoff);\n"
             "                if (pos + 3 < 25) aq[(pp + 1) & 1][1] = ldv(vb + (pos + 3) * W24_PS + aoff); }")
    s = _sub(s, "            W24_STAMP(chunk, 4);\n            __syncthreads();", "            W24_STAMP(chunk, 4);\n            if (!(ABL & 8)) __syncthreads();")
    s = _sub(s, "    for (int chunk0 = 0; chunk0 < nch; chunk0 += 2) {", "    for (int chunk0 = 0; chunk0 < ((ABL & 128) ? 0 : nch); chunk0 += 2) {")
    s = _sub(s, "    // ---- output transform A^T M A (A^T = [1 1 1 1 0; 0 1 -1 2 1]) of the lane's 4 tiles x 1 channel, fused epilogue.",
             "    if (ABL & 64) { float t = 0.f; for (int i = 0; i < 25; ++i) for (int r = 0; r < 4; ++r) t += acc[i][r]; if (t == 123.456f) p.out[0] = t; return; }\n"
             "    // ---- output transform A^T M A (A^T = [1 1 1 1 0; 0 1 -1 2 1]) of the lane's 4 tiles x 1 channel, fused epilogue.")
    return s


def build():
    sys.path.insert(0, ROOT)
    from rpg_ramnet_amd import build as B
    B.build()                                               # the objects of the unpatched sources
    src = os.path.join(SCRATCH, "pkg", "csrc")
    shutil.rmtree(SCRATCH, ignore_errors=True)
    os.makedirs(os.path.join(SCRATCH, "include"))
    shutil.copytree(B.CSRC, src)
    shutil.copy(os.path.join(ROOT, "include", "ramnet_hip.h"), os.path.join(SCRATCH, "include"))
    for stem, fn in (("conv_wgrad_wino", patch_wgrad), ("conv_wino", patch_wino), ("conv_wino24", patch_wino24), ("conv_wino6", patch_wino6)):
        if stem not in MASKS:
            continue
        p = os.path.join(src, stem + ".hip")
        with open(p) as f:
            s = f.read()
        with open(p, "w") as f:
            f.write(fn(s))
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def one(job):
        stem, m = job
        obj = os.path.join(SCRATCH, "%s_%d.o" % (stem, m))
        subprocess.run([hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(stem + ".hip", []) + ["-DABL=%d" % m, "-c", os.path.join(src, stem + ".hip"), "-o", obj], check=True)
        others = [os.path.join(B.OBJ, f) for f in os.listdir(B.OBJ) if f.endswith(".o") and f != stem + ".o"]
        lib = os.path.join(OUT, "%s_%d.so" % (stem, m))
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + others + [obj], check=True)
        return lib

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for lib in ex.map(one, [(s, m) for s, ms in MASKS.items() for m in ms]):
            print("built", lib)


def run():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ablation.txt"), "w") as out:
        for stem, ms in MASKS.items():
            for m in ms:
                lib = os.path.join(OUT, "%s_%d.so" % (stem, m))
                if not os.path.exists(lib):
                    continue
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_layers.py"), "--only", ONLY[stem], "--reps", "20"],
                                   env=dict(os.environ, RAMNET_HIP_LIB=lib), capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("sum")]
                f = line[0].split() if line else []
                msg = "%-16s ABL=%-3d  %s: forward %s  backward-data %s  backward-weights %s ms" % (
                    stem, m, "six ConvGRU launches" if ONLY[stem] == "gru" else "three decoders (+ plain variants)",
                    f[3] if f else "?", f[6] if f else "?", f[9] if f else "?")
                print(msg)
                out.write(msg + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 2:                                   # restrict to some kernels: build conv_wino24
        MASKS = {k: v for k, v in MASKS.items() if k in sys.argv[2:]}
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "build"]()
