"""CPU: guards on the gfx950 code the compiler actually emitted (csrc/build.sh keeps every object's device assembly).

1. `v_pk_fma_f32` with a HIGH-half broadcast of one source -- `op_sel:[0,1,0]` and its twins: both packed results read the high half of
   the same register pair -- returned wrong sums in lanes 48..63 while ANOTHER PROCESS ran MFMA kernels on the same GPU
   (profiles/r03_pkfma_cross_process.txt, scripts/probes/pkfma_probe.hip).  The embed kernel was rewritten so that
   the compiler no longer picks that form; nothing in the language pins code generation, so this test does: it fails, naming the
   kernels, if the form comes back in any object of the library.  The forms that stayed bit-equal beside a neighbour process --
   low-half broadcasts (`op_sel_hi` bit cleared with `op_sel` bit clear) and half swaps -- are listed, not failed.
2. No kernel of the library spills registers to scratch (a spill in a K loop is a 2-10x slowdown that no test of results would see).
"""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "video-to-audio-and-piano-rp_amd", "csrc", "build")
SOURCES = ("gemm", "gemm_8phase", "rowops", "attention", "conv", "vocoder", "qproj_xattn")


@pytest.fixture(scope="module")
def listings():
    from v2a_amd import _lib
    _lib.build(verbose=False)               # no-op when the objects are newer than the sources
    files = {os.path.basename(f).split("-hip-")[0]: f for f in glob.glob(os.path.join(BUILD, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))}
    missing = [s for s in SOURCES if s not in files and os.path.exists(os.path.join(os.path.dirname(BUILD), s + ".hip"))]
    assert not missing, "no device assembly for %s: csrc/build.sh must compile with -save-temps=obj" % missing
    return files


def _vec(text, name, default):
    m = re.search(name + r":\[([01]),([01]),([01])\]", text)
    return [int(v) for v in m.groups()] if m else list(default)


def packed_fma_forms(path):
    """{kernel: {(op_sel, op_sel_hi): count}} of the v_pk_fma_f32 instructions of one listing."""
    out, kern = {}, None
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):\s", line)
        if m and not line.startswith(".L"):
            kern = m.group(1)
        if "v_pk_fma_f32" in line:
            sel, hi = _vec(line, "op_sel", (0, 0, 0)), _vec(line, "op_sel_hi", (1, 1, 1))
            out.setdefault(kern, {}).setdefault((tuple(sel), tuple(hi)), 0)
            out[kern][(tuple(sel), tuple(hi))] += 1
    return out


def test_no_high_half_broadcast_in_packed_fma(listings):
    bad, other = [], {}
    for src, path in sorted(listings.items()):
        for kern, forms in packed_fma_forms(path).items():
            for (sel, hi), n in forms.items():
                if any(s == 1 and h == 1 for s, h in zip(sel, hi)):       # both packed results read the HIGH half of that source
                    bad.append("%s.hip: %s: %d x v_pk_fma_f32 op_sel:%s op_sel_hi:%s" % (src, kern, n, list(sel), list(hi)))
                elif sel != (0, 0, 0) or hi != (1, 1, 1):
                    other[(sel, hi)] = other.get((sel, hi), 0) + n
    print("packed-FMA modifier forms kept (low-half broadcasts / half swaps):", {("%s/%s" % k): v for k, v in sorted(other.items())})
    assert not bad, "high-half broadcast forms of v_pk_fma_f32 (wrong sums beside another process's MFMA kernels):\n" + "\n".join(bad)


# the 256x256 tile on the plain ring (2 stages of 64 KB, 128 accumulators per lane): the fallback for wide outputs when the 8-phase kernel
# is switched off through v2a_set_tuning (A/B only) -- 12-22 spilled registers in its prologue / epilogue, never on the sampler's path
KNOWN_SPILLS = ("gemm_bf16_dma_kernelILi0EfLi256ELi256E", "gemm_bf16_dma_kernelILi0EDF16bLi256ELi256E", "gemm_bf16_dma_kernelILi2EfLi256ELi256E",
                "gemm_bf16_dma_kernelILi2EDF16bLi256ELi256E", "gemm_bf16_dma_kernelILi3EfLi256ELi256E", "gemm_bf16_dma_kernelILi4EfLi256ELi256E")


# per-kernel scratch allowances (bytes per lane): none -- no kernel on any path but KNOWN_SPILLS may touch scratch
SCRATCH_ALLOWED = {}


def test_no_scratch_spills(listings):
    spilled = []
    for src, path in sorted(listings.items()):
        text = open(path).read()
        # the metadata block lists, per kernel, .private_segment_fixed_size / .sgpr_spill_count / .vgpr_spill_count in alphabetical key order
        for blk in text.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            vs = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
            ps = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if name and any(k in name.group(1) for k in KNOWN_SPILLS):
                continue
            allowed = max([v for k, v in SCRATCH_ALLOWED.items() if name and k in name.group(1)] or [0])
            if name and ps and int(ps.group(1)) > allowed:
                spilled.append("%s.hip: %s: %s VGPRs spilled, %s B scratch (allowed %d)" % (src, name.group(1)[:90], vs.group(1) if vs else "?", ps.group(1), allowed))
    assert not spilled, "kernels with scratch:\n" + "\n".join(spilled)


def test_no_scratch_access_inside_mfma_loops(listings):
    """Whatever a kernel spills, nothing is stored to or reloaded from scratch between its first and its last MFMA instruction (the K loop)."""
    bad = []
    for src, path in sorted(listings.items()):
        kern, first, last, hits = None, None, None, []
        def close():
            if kern and first is not None and not any(k in kern for k in KNOWN_SPILLS):
                inside = [n for n in hits if first < n < last]
                if inside:
                    bad.append("%s.hip: %s: %d scratch accesses inside the MFMA loop" % (src, kern[:90], len(inside)))
        for n, line in enumerate(open(path)):
            m = re.match(r"^(_Z[\w.$]*):", line)
            if m:
                close()
                kern, first, last, hits = m.group(1), None, None, []
            elif "v_mfma" in line:
                first = n if first is None else first
                last = n
            elif "scratch_" in line and ("scratch_load" in line or "scratch_store" in line):
                hits.append(n)
        close()
    assert not bad, "\n".join(bad)
