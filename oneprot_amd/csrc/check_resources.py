#!/usr/bin/env python3
"""Reads hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin or files) and fails when a PRODUCT kernel instantiation carries
scratch (= a register spill: a reload queues behind the epilogue's store burst, oneprot_amd/csrc/gemm_nt8.hip) beyond what the table below
records as known.  build.sh compiles the hot translation units with the remark enabled and pipes it through here, so that a spill regression
fails the build instead of showing up as a slower bench.   usage: check_resources.py [--list] remarks.txt ..."""
import re
import subprocess
import sys

# demangled-name regex -> bytes/lane of scratch tolerated (0 = must be spill-free)
RULES = [
    (r"^void g8::k_gemm8<", 0),                 # every 8-phase NT GEMM instantiation (FFN-1 with one or two outputs, QKV + RoPE, dgrads, ...)
    (r"^void k_gemm_tn8<", 0),                  # 8-phase weight-gradient GEMM
    (r"^void gln::k_gemm_ln", 0),               # out-projection + residual + LayerNorm
    (r"^void k_attn_fwd3<", 0),                 # persistent attention forward
    (r"^void k_attn_fwd3w<", 0),                # ... for 128-byte rows (hd 64): 256 registers; the short per-tile list of k_attn_fwd3 (round 6) spilled 20 bytes here and was not taken over
    (r"^void k_attn_fwd2<", 16),                # (long-sequence / hd 64 path: the output address computed at entry is parked in scratch until the final store -- outside every loop)
    (r"^void k_attn_bwd_fused<\(int\)32>", 0),
    (r"^void k_attn_bwd_fused64<\(int\)32>", 0),   # 8-wave fused backward (two key blocks per wave): 254 of 256 registers, see the kernel's comments before adding a live value
    (r"^void k_layernorm_(fwd|bwd)", 0),
]


def parse(text):
    out = {}
    cur = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                out[cur][key] = int(m.group(1))
    return out


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, p.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    text = "".join(open(a).read() for a in args) if args else sys.stdin.read()
    kernels = parse(text)
    names = demangle(list(kernels))
    bad = []
    for mangled, res in sorted(kernels.items()):
        dn = names[mangled]
        limit = next((lim for pat, lim in RULES if re.search(pat, dn)), None)
        if "--list" in sys.argv:
            print(f"{res.get('vgpr', '?'):>4} VGPR {res.get('scratch', '?'):>4} B scratch  occ {res.get('occ', '?')}  {dn[:150]}")
        if limit is not None and res.get("scratch", 0) > limit:
            bad.append((dn, res))
    for dn, res in bad:
        print(f"SPILL: {res['scratch']} bytes/lane of scratch, {res.get('vgpr')} VGPRs: {dn[:200]}", file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
