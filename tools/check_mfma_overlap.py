#!/usr/bin/env python3
"""Scan the gfx950 assembly of the HIP sources for MFMA instructions whose destination tuple PARTIALLY overlaps
their accumulator input (D != C but sharing registers).  ROCm 7.2 emits these for v_mfma_f32_16x16x32_bf16 builtins;
on MI355X they produced timing-dependent wrong results in the conv phase of the training kernel (DESIGN.md,
"compiler hazard").  The bf16x3 path therefore issues that instruction as tied inline asm; this script is the guard
that no builtin-generated instance is left.  Usage: tools/check_mfma_overlap.py [file.hip ...]   (exit 1 on a hit)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "purejaxql_amd", "csrc")
PAT = re.compile(r'(v_mfma_\w+)\s+([av])\[(\d+):(\d+)\],\s*[av]\[\d+:\d+\],\s*[av]\[\d+:\d+\],\s*([av])\[(\d+):(\d+)\]')


def scan(asm_path):
    hits, cur = [], "?"
    for ln, line in enumerate(open(asm_path), 1):
        if line.startswith("_Z") and ":" in line:
            cur = line.split(":")[0]
        m = PAT.search(line)
        if not m:
            continue
        dk, d0, d1, ck, c0, c1 = m.group(2), int(m.group(3)), int(m.group(4)), m.group(5), int(m.group(6)), int(m.group(7))
        if dk == ck and (d0, d1) != (c0, c1) and not (d1 < c0 or c1 < d0):
            hits.append((cur, ln, line.strip()))
    return hits


M0_OK = re.compile(r'^\s*s_mov_b32 m0, s\d+\s*$')


def scan_m0(asm_path):
    """pqn_qnet_pos.hip leaves M0 holding the LDS-DMA destination (pos_dma16 does not save / restore it since round 6): every
    mention of m0 in its assembly must be one of those writes -- nothing may READ it (s_movrel, sendmsg, GWS, v_readlane m0 ...)."""
    return [(ln, line.strip()) for ln, line in enumerate(open(asm_path), 1)
            if re.search(r'\bm0\b', line.split(";")[0]) and not M0_OK.match(line.split(";")[0])]


def main():
    files = [os.path.abspath(f) for f in sys.argv[1:]] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "x.s")
            extra = ["-fno-slp-vectorize"] if os.path.basename(f) == "pqn_qnet_pos.hip" else []     # as the Makefile builds it
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", *extra,
                            "--cuda-device-only", "-S", f, "-o", out], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
            hits = scan(out)
            if os.path.basename(f) == "pqn_qnet_pos.hip":
                m0 = scan_m0(out)
                print("%-20s %d uses of M0 other than the LDS-DMA destination writes" % (os.path.basename(f), len(m0)))
                for ln, l in m0[:8]:
                    print("    line %d  %s" % (ln, l))
                bad += len(m0)
        print("%-20s %d partially overlapping MFMA D/C" % (os.path.basename(f), len(hits)))
        for k, ln, l in hits[:12]:
            print("    %s:%d  %s" % (k[:50], ln, l))
        bad += len(hits)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
