"""ctypes binding of libv2a_cfm.so (the C ABI declared in include/v2a_cfm.h).

The library is built in-tree by csrc/build.sh (hipcc --offload-arch=gfx950) and loaded
AFTER `import torch`, so its libamdhip64.so.7 dependency resolves to the HIP runtime torch
already loaded and torch streams / device pointers are valid in it.  There is no CPU
fallback: if the library is missing, every op raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libv2a_cfm.so")      # in-tree, next to this file; nothing is read from the environment
CSRC = os.path.join(_HERE, "csrc")

F32, BF16, BF16_SPLIT = 0, 1, 2
EPI_STORE, EPI_SIGMOID, EPI_GEGLU, EPI_RESID, EPI_GATE_RESID = 0, 1, 2, 3, 4

_lib = None


class V2AError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p * 3), ("lda", C.c_int64 * 3), ("ka", C.c_int32 * 3),
        ("nseg", C.c_int32), ("a_dtype", C.c_int32),
        ("w", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("compute_dtype", C.c_int32), ("epilogue", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_dtype", C.c_int32),
        ("out_bf16", C.c_void_p), ("ld_out_bf16", C.c_int64),
        ("resid", C.c_void_p), ("ldr", C.c_int64),
        ("gate", C.c_void_p), ("step", C.c_void_p),
        ("gate_step_stride", C.c_int64), ("gate_batch_stride", C.c_int64), ("rows_per_batch", C.c_int32),
        ("rope_table", C.c_void_p), ("rope_cols", C.c_int32), ("rope_pos_offset", C.c_int32), ("relu", C.c_int32),
        ("a_row_offset", C.c_void_p), ("a_ktile_offset", C.c_void_p), ("out_row_offset", C.c_void_p),
        ("tile_hint", C.c_int32),
        ("norm_gamma", C.c_void_p), ("norm_step_stride", C.c_int64), ("norm_batch_stride", C.c_int64),
        ("norm_switch_row", C.c_int32), ("norm_switch_offset", C.c_int32),
        ("norm_ssq", C.c_void_p), ("ld_norm_ssq", C.c_int64),
        ("row_ssq", C.c_void_p), ("ld_row_ssq", C.c_int64), ("row_ssq_parts", C.c_int32), ("row_norm_dim", C.c_int32),
        ("out_bf16_split", C.c_int32),
        ("a_lo_offset", C.c_int64 * 3), ("out_bf16_lo_offset", C.c_int64),
    ]


class DwconvNorm(C.Structure):
    """Mirror of `v2a_dwconv_norm`: the RMSNorm after the depthwise convolution, folded into it."""
    _fields_ = [("out_bf16", C.c_void_p), ("ld_out_bf16", C.c_int64), ("norm_gamma", C.c_void_p), ("step", C.c_void_p),
                ("norm_step_stride", C.c_int64), ("norm_batch_stride", C.c_int64), ("norm_ssq", C.c_void_p), ("ld_norm_ssq", C.c_int64),
                ("split", C.c_int32), ("reserved", C.c_int32)]


class Tuning(C.Structure):
    """Mirror of `v2a_tuning` (include/v2a_cfm.h): explicit tile-selection overrides, nothing is read from the environment."""
    _fields_ = [("gemm_force_tile", C.c_int32), ("gemm_k_rotation", C.c_int32), ("gemm_8phase", C.c_int32),
                ("gemm_8phase_min_tiles", C.c_int32), ("dwconv_rows_per_wave", C.c_int32), ("gemm_xcd_order_1x8", C.c_int32),
                ("attn_one_group_from", C.c_int32), ("reserved", C.c_int32 * 1)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("gate", C.c_void_p), ("out", C.c_void_p),
        ("q_row_stride", C.c_int64), ("k_row_stride", C.c_int64), ("v_row_stride", C.c_int64),
        ("gate_row_stride", C.c_int64), ("out_row_stride", C.c_int64),
        ("q_batch_stride", C.c_int64), ("k_batch_stride", C.c_int64), ("v_batch_stride", C.c_int64),
        ("gate_batch_stride", C.c_int64), ("out_batch_stride", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32),
        ("kv_len", C.c_void_p), ("q_len", C.c_void_p),
        ("scale", C.c_float), ("softclamp", C.c_float), ("dtype", C.c_int32), ("out_split", C.c_int32),
    ]


class RollHeadArgs(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("x2", "x3", "x4", "x5")] + [("B", C.c_int32), ("P", C.c_int32)] +
                [(f"frb{i}_{n}", C.c_void_p) for i in (4, 3, 2) for n in ("w1t", "b1", "w2t", "b2")] +
                [(n, C.c_void_p) for n in ("conv2_wt", "conv2_b", "fc_wt", "fc_b")] +
                [("notes", C.c_int32), ("apply_sigmoid", C.c_int32), ("out", C.c_void_p)])


EXPORTS = [
    "v2a_abi_version", "v2a_last_error", "v2a_gemm", "v2a_gemm_args_size", "v2a_set_tuning", "v2a_rmsnorm", "v2a_dwconv_silu_residual", "v2a_dwconv_silu_residual_norm",
    "v2a_rope_inplace", "v2a_attention", "v2a_qproj_xattn", "v2a_linear_small", "v2a_fill_registers", "v2a_time_cond",
    "v2a_apg_reduce", "v2a_cfg_euler", "v2a_step_advance", "v2a_cast_bf16", "v2a_split_bf16",
    "v2a_im2col", "v2a_frames_pack", "v2a_pool2d", "v2a_roll_head", "v2a_roll_expand",
    "v2a_elu_pad", "v2a_lstm_layer", "v2a_lstm2",
]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libv2a_cfm.so (cross-compiles without a GPU)."""
    if force:
        for f in os.listdir(os.path.join(CSRC, "build")) if os.path.isdir(os.path.join(CSRC, "build")) else []:
            os.remove(os.path.join(CSRC, "build", f))
    r = subprocess.run(["bash", os.path.join(CSRC, "build.sh")], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode != 0:
        raise V2AError("hipcc build of libv2a_cfm.so failed")
    return LIB_PATH


ABI_VERSION = 8          # what v2a_abi_version() of this source tree returns (csrc/rowops.hip)


def _declare(lib):
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.v2a_abi_version.restype = C.c_int
    lib.v2a_last_error.restype = C.c_char_p
    lib.v2a_gemm.argtypes = [C.POINTER(GemmArgs), vp]
    lib.v2a_gemm_args_size.restype = C.c_int
    lib.v2a_set_tuning.argtypes = [C.POINTER(Tuning)]
    lib.v2a_attention.argtypes = [C.POINTER(AttnArgs), vp]
    lib.v2a_qproj_xattn.argtypes = [C.POINTER(GemmArgs), C.POINTER(AttnArgs), vp]
    lib.v2a_rmsnorm.argtypes = [vp, i64, vp, i64, i32, i64, i32, vp, vp, i64, i64, i32, vp]
    lib.v2a_dwconv_silu_residual.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.v2a_dwconv_silu_residual_norm.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, C.POINTER(DwconvNorm), vp]
    lib.v2a_rope_inplace.argtypes = [vp, i32, i64, i64, i32, i32, i32, vp, i32, vp]
    lib.v2a_linear_small.argtypes = [vp, i64, i32, vp, vp, vp, i32, vp, i64, i32, i32, i32, vp, vp, vp]
    lib.v2a_fill_registers.argtypes = [vp, i64, vp, i32, i32, i32, vp]
    lib.v2a_time_cond.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp]
    lib.v2a_apg_reduce.argtypes = [vp, vp, i32, i32, i32, i64, i32, vp, vp]
    lib.v2a_cfg_euler.argtypes = [vp, vp, i32, i32, i32, i64, i32, f32, vp, vp, vp, f32, vp]
    lib.v2a_step_advance.argtypes = [vp, vp]
    lib.v2a_cast_bf16.argtypes = [vp, vp, i64, vp]
    lib.v2a_split_bf16.argtypes = [vp, i64, vp, i64, i64, i32, vp]
    lib.v2a_im2col.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, i32, i32, vp]
    lib.v2a_frames_pack.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.v2a_pool2d.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.v2a_roll_head.argtypes = [C.POINTER(RollHeadArgs), vp]
    lib.v2a_roll_expand.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.v2a_elu_pad.argtypes = [vp, vp, i64, i32, i32, i32, i32, vp]
    lib.v2a_lstm_layer.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp]
    lib.v2a_lstm2.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    for name in EXPORTS:
        if name not in ("v2a_abi_version", "v2a_last_error", "v2a_gemm_args_size"):
            getattr(lib, name).restype = C.c_int


def lib():
    """The loaded library; raises loudly if it was never built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise V2AError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the sampler.")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
        if _lib.v2a_abi_version() != ABI_VERSION:
            raise V2AError("libv2a_cfm.so reports ABI version %d, this binding was written for %d: rebuild (csrc/build.sh)"
                           % (_lib.v2a_abi_version(), ABI_VERSION))
        if _lib.v2a_gemm_args_size() != C.sizeof(GemmArgs):
            raise V2AError("libv2a_cfm.so was built with sizeof(v2a_gemm_args) = %d, this binding mirrors %d bytes: rebuild "
                           "(csrc/build.sh)" % (_lib.v2a_gemm_args_size(), C.sizeof(GemmArgs)))
    return _lib


def set_tuning(force_tile: int = -1, k_rotation: bool = False, eight_phase: int | None = None, eight_phase_min_tiles: int = 0,
               dwconv_rows_per_wave: int = 0, xcd_order_1x8: bool = False, attn_one_group_from: int = 0, reserved: int = 0):
    """Tile-selection overrides (A/B measurements); `set_tuning()` restores the library defaults.  `reserved`: probe builds only."""
    if (force_tile == -1 and not k_rotation and eight_phase is None and eight_phase_min_tiles == 0 and dwconv_rows_per_wave == 0
            and not xcd_order_1x8 and attn_one_group_from == 0 and not reserved):
        check(lib().v2a_set_tuning(None))
        return
    t = Tuning(force_tile, 1 if k_rotation else 0, 1 if eight_phase is None else eight_phase, eight_phase_min_tiles, dwconv_rows_per_wave,
               1 if xcd_order_1x8 else 0, attn_one_group_from, (C.c_int32 * 1)(reserved))
    check(lib().v2a_set_tuning(C.byref(t)))


def check(rc: int):
    if rc != 0:
        raise V2AError(f"v2a_cfm call failed ({rc}): {lib().v2a_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dt_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise V2AError(f"unsupported dtype {t}")


def _p(t):
    return 0 if t is None else t.data_ptr()


# ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream ----
_prof = None


class KernelProfiler:
    """Per-launch timing of the C-ABI calls with HIP events on the stream the kernel is launched on (torch's current
    stream), aggregated by kernel instantiation.

    external=False : eager -- every launch sits between its own pair of events (`inner` back-to-back launches per pair for
                     GEMMs amortise the ~5 us dispatch gap an eager pair includes, at the price of warm caches).
    external=True  : for use under hipGraph capture -- ONE external event is recorded in front of every launch (an event
                     record node of the graph) and one after the last; after each replay `collect()` adds the interval to
                     the next mark to the launch's total.  Intervals are kernel time + the launch boundary behind it, with
                     every kernel launched exactly once per evaluation in its real cache state."""

    def __init__(self, shapes: bool = False, inner: int = 1, external: bool = False):
        self.recs = []
        self.shapes = shapes      # key GEMM launches by MxNxK as well
        self.inner = inner
        self.external = external
        self.marks, self.end, self.tot = [], None, {}

    def launch(self, key, flops, nbytes, call):
        if self.external:
            e = torch.cuda.Event(enable_timing=True, external=True)
            e.record()
            self.marks.append((key, flops, nbytes, e))
            call()
            return
        n = self.inner if key.startswith("gemm") else 1
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            call()
        e.record()
        self.recs.append((key, flops, nbytes, s, e, n))

    def close(self):
        """external mode: the mark behind the last launch (call inside the capture)."""
        self.end = torch.cuda.Event(enable_timing=True, external=True)
        self.end.record()

    def collect(self):
        """external mode: after a replay has completed, add this replay's intervals."""
        for i, (key, flops, nbytes, e) in enumerate(self.marks):
            nxt = self.marks[i + 1][3] if i + 1 < len(self.marks) else self.end
            a = self.tot.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += e.elapsed_time(nxt)
            a["flops"] += flops
            a["bytes"] += nbytes

    def summary(self):
        torch.cuda.synchronize()
        if self.external:
            return self.tot
        agg = {}
        for key, flops, nbytes, s, e, n in self.recs:
            a = agg.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e) / n
            a["flops"] += flops
            a["bytes"] += nbytes
        return agg


def set_profiler(p):
    global _prof
    _prof = p


def _launch(key, flops, nbytes, call):
    if _prof is None:
        check(call())
    else:
        _prof.launch(key, flops, nbytes, lambda: check(call()))


_EPI_NAMES = {0: "store", 1: "sigmoid", 2: "geglu", 3: "resid", 4: "gate_resid"}


# ------------------------------------------------------------------------------------------
# thin typed wrappers (torch tensors are only carriers of device pointers here)
# ------------------------------------------------------------------------------------------

def gemm(a_segs, w, out, **kw):
    """One v2a_gemm launch; arguments as for gemm_args."""
    g, key, flops, nbytes = gemm_args(a_segs, w, out, **kw)
    _launch(key, flops, nbytes, lambda: lib().v2a_gemm(C.byref(g), stream_ptr()))


def gemm_args(a_segs, w, out, *, M, N, compute, epilogue=EPI_STORE, bias=None, resid=None, gate=None,
         step=None, gate_step_stride=0, gate_batch_stride=0, rows_per_batch=0, ldo=None, ldr=None,
         out_bf16=None, ld_out_bf16=None, rope_table=None, rope_cols=0, rope_pos_offset=0, relu=False,
         a_row_offset=None, a_ktile_offset=None, out_row_offset=None, tile_hint=0,
         norm_gamma=None, norm_step_stride=0, norm_batch_stride=0, norm_switch_row=0, norm_switch_offset=0, norm_ssq=None,
         row_ssq=None, row_norm_dim=0, out_bf16_split=False, a_split=False, out_split=False, out_bf16_lo_offset=0):
    """a_segs: list of (tensor_or_ptr_view, lda, k).  w: [N][K] tensor in the compute dtype.
    a_split: the segments are V2A_BF16_SPLIT rows ([hi k | lo k], lda >= 2k; a segment may be a 4-tuple whose last element is the offset of its
    lo plane when that is not k: v2a_gemm_args.a_lo_offset) and w is [N][2K] = [W_hi | W_lo] (the bf16x3 mode's native
    GEMM: three MFMA products per fp32 product); out_split: GEGLU output as hi | lo planes; out_bf16_split: the shadow likewise.
    Folded RMSNorm (v2a_gemm_args): producer -- norm_gamma (+ strides / switch) scales the out_bf16 shadow, norm_ssq (rows, N/32)
    receives the sums of squares; consumer -- row_ssq (rows, parts) of its A rows and row_norm_dim = their width."""
    g = GemmArgs()
    for i, seg in enumerate(a_segs):
        t, lda, k = seg[:3]
        g.a[i] = t.data_ptr()
        g.lda[i] = lda
        g.ka[i] = k
        g.a_lo_offset[i] = seg[3] if len(seg) > 3 else 0
    a_segs = [seg[:3] for seg in a_segs]
    g.nseg = len(a_segs)
    g.a_dtype = BF16_SPLIT if a_split else dt_code(a_segs[0][0].dtype)
    g.w = w.data_ptr()
    g.ldw = w.stride(0)
    g.bias = _p(bias)
    g.M, g.N = M, N
    g.compute_dtype = compute
    g.epilogue = epilogue
    g.out = out.data_ptr()
    g.ldo = ldo if ldo is not None else out.stride(-2)
    g.out_dtype = BF16_SPLIT if out_split else dt_code(out.dtype)
    g.out_bf16 = _p(out_bf16)
    g.ld_out_bf16 = (ld_out_bf16 if ld_out_bf16 is not None else g.ldo) if out_bf16 is not None else 0
    g.resid = _p(resid)
    g.ldr = (ldr if ldr is not None else (resid.stride(-2) if resid is not None else 0))
    g.gate = _p(gate)
    g.step = _p(step)
    g.gate_step_stride = gate_step_stride
    g.gate_batch_stride = gate_batch_stride
    g.rows_per_batch = rows_per_batch
    g.rope_table = _p(rope_table)
    g.rope_cols, g.rope_pos_offset = rope_cols, rope_pos_offset
    g.relu = 1 if relu else 0
    g.a_row_offset, g.a_ktile_offset, g.out_row_offset = _p(a_row_offset), _p(a_ktile_offset), _p(out_row_offset)
    g.tile_hint = tile_hint if compute == BF16 and g.a_dtype in (BF16, BF16_SPLIT) and epilogue != EPI_SIGMOID else 0
    g.norm_gamma = _p(norm_gamma)
    g.norm_step_stride, g.norm_batch_stride = norm_step_stride, norm_batch_stride
    g.norm_switch_row, g.norm_switch_offset = norm_switch_row, norm_switch_offset
    g.norm_ssq = _p(norm_ssq)
    g.ld_norm_ssq = norm_ssq.stride(-2) if norm_ssq is not None else 0
    g.row_ssq = _p(row_ssq)
    g.ld_row_ssq = row_ssq.stride(-2) if row_ssq is not None else 0
    g.row_ssq_parts = row_norm_dim // 32 if row_ssq is not None else 0
    g.row_norm_dim = row_norm_dim
    g.out_bf16_split = 1 if out_bf16_split else 0
    g.out_bf16_lo_offset = out_bf16_lo_offset
    K = sum(k for _, _, k in a_segs)
    key = "gemm<%s,%s,%s,%s>" % ("bf16" if compute == BF16 else "f32", "a_f32" if g.a_dtype == F32 else ("a_split" if a_split else "a_bf16"),
                                 _EPI_NAMES[epilogue], "f32" if g.out_dtype == F32 else "bf16")
    esz = 2 if compute == BF16 else 4
    nbytes = M * K * (4 if g.a_dtype == F32 else 2) + N * K * esz + M * (N // 2 if epilogue == EPI_GEGLU else N) * out.element_size()
    nbytes += (M * N * 4 if resid is not None else 0) + (M * N * 2 if out_bf16 is not None else 0)      # residual rows read, bf16 shadow written
    if g.tile_hint:
        key = key[:-1] + ",tile%d>" % (g.tile_hint - 1)        # side-stream launches: tile shape chosen for running beside others
    if _prof is not None and _prof.shapes:
        key += " %dx%dx%d" % (M, N, K)
    return g, key, (6.0 if a_split else 2.0) * M * N * K, nbytes * (2 if a_split else 1)


def rmsnorm(x, y, *, rows, d, gamma, step=None, gamma_step_stride=0, gamma_batch_stride=0, rows_per_batch=0,
            ldx=None, ldy=None, split=False):
    """split=True: y is a (rows, 2*d) bf16 buffer that receives the hi | lo planes (BF16_SPLIT)."""
    ydt = BF16_SPLIT if split else dt_code(y.dtype)
    if split:
        ldy = ldy or 2 * d
    _launch("rmsnorm<%s>" % ("f32" if y.dtype == torch.float32 else ("bf16x2" if split else "bf16")), 0.0,
            rows * d * (4 + y.element_size() * (2 if split else 1)),
            lambda: lib().v2a_rmsnorm(x.data_ptr(), ldx or d, y.data_ptr(), ldy or d, ydt, rows, d,
                                      gamma.data_ptr(), _p(step), gamma_step_stride, gamma_batch_stride, rows_per_batch,
                                      stream_ptr()))


def _dwconv_norm(norm, d):
    n = DwconvNorm()
    n.out_bf16, n.ld_out_bf16 = norm["out_bf16"].data_ptr(), norm.get("ld_out_bf16", d)
    n.norm_gamma, n.step = norm["gamma"].data_ptr(), _p(norm.get("step"))
    n.norm_step_stride, n.norm_batch_stride = norm.get("step_stride", 0), norm.get("batch_stride", 0)
    n.norm_ssq, n.ld_norm_ssq = norm["ssq"].data_ptr(), norm["ssq"].stride(-2)
    n.split = 1 if norm.get("split") else 0
    if n.split:
        n.ld_out_bf16 = norm.get("ld_out_bf16", 2 * d)
    return n


def dwconv(x, out, wt, bias, *, B, N, d, ksize, lens=None, norm=None):
    """norm = dict(out_bf16, gamma, ssq, step=None, step_stride=0, batch_stride=0): the RMSNorm after the conv folded in."""
    if norm is None:
        _launch("dwconv", 2.0 * B * N * d * ksize, B * N * d * 8,
                lambda: lib().v2a_dwconv_silu_residual(x.data_ptr(), out.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                                                       B, N, d, ksize, _p(lens), stream_ptr()))
        return
    n = _dwconv_norm(norm, d)
    # x read + out written (fp32) + the normalised bf16 copy: one plane, or hi | lo planes in the split mode
    _launch("dwconv+norm", 2.0 * B * N * d * ksize, B * N * d * (12 if n.split else 10),
            lambda: lib().v2a_dwconv_silu_residual_norm(x.data_ptr(), out.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                                                        B, N, d, ksize, _p(lens), C.byref(n), stream_ptr()))


def rope(qk, *, rows, row_stride, nheads, rows_per_batch, pos_offset, table, layout):
    _launch("rope<%s>" % ("f32" if qk.dtype == torch.float32 else "bf16"), 0.0, rows * nheads * 64 * 2 * qk.element_size(),
            lambda: lib().v2a_rope_inplace(qk.data_ptr(), dt_code(qk.dtype), rows, row_stride, nheads, rows_per_batch,
                                           pos_offset, table.data_ptr(), layout, stream_ptr()))


def attention(q, k, v, gate, out, **kw):
    """q,k,v,gate,out: integer device addresses (views into fused buffers); one v2a_attention launch."""
    a, key, flops, nbytes = attention_args(q, k, v, gate, out, **kw)
    _launch(key, flops, nbytes, lambda: lib().v2a_attention(C.byref(a), stream_ptr()))


def attention_args(q, k, v, gate, out, *, strides, B, H, Nq, Nk, kv_len=None, q_len=None, scale, softclamp, dtype, out_split=False):
    a = AttnArgs()
    a.q, a.k, a.v, a.gate, a.out = q, k, v, gate, out
    (a.q_row_stride, a.k_row_stride, a.v_row_stride, a.gate_row_stride, a.out_row_stride,
     a.q_batch_stride, a.k_batch_stride, a.v_batch_stride, a.gate_batch_stride, a.out_batch_stride) = strides
    a.B, a.H, a.Nq, a.Nk = B, H, Nq, Nk
    a.kv_len, a.q_len = _p(kv_len), _p(q_len)
    a.scale, a.softclamp, a.dtype = scale, softclamp, dtype
    a.out_split = 1 if out_split else 0
    esz = 2 if dtype == BF16 else 4
    return a, "attention<%s>" % {BF16: "bf16", F32: "f32", BF16_SPLIT: "bf16x3"}[dtype], 4.0 * B * H * Nq * Nk * 64, B * H * 64 * (2 * Nq + 2 * Nk) * esz


def qproj_xattn(a, lda, K, w, *, bias, M, N, rows_per_batch, k, v, out, kv_strides, out_strides, B, H, Nk, kv_len=None, q_len=None,
                scale, softclamp, rope_table=None, rope_cols=0, rope_pos_offset=0, row_ssq=None, row_norm_dim=0, split=False):
    """q-projection + cross-attention in one launch (v2a_qproj_xattn).  a: (M, K) bf16 rows; w: [N >= H*65][K] = to_q rows then the
    gate rows; k, v, out: integer device addresses; kv_strides = (k_row, v_row, k_batch, v_batch), out_strides = (row, batch).
    split=True (bf16x3 mode): a rows = [hi K | lo K], w rows = [W_hi | W_lo], k / v fp32, out rows = hi | lo planes (2 * H * 64 bf16)."""
    g = GemmArgs()
    g.a[0], g.lda[0], g.ka[0], g.nseg = a.data_ptr(), lda, K, 1
    g.compute_dtype = BF16
    g.a_dtype = BF16_SPLIT if split else BF16
    g.w, g.ldw = w.data_ptr(), w.stride(0)
    g.bias = _p(bias)
    g.M, g.N = M, N
    g.epilogue = EPI_STORE
    g.out_dtype = F32 if split else BF16
    g.rows_per_batch = rows_per_batch
    g.rope_table = _p(rope_table)
    g.rope_cols, g.rope_pos_offset = rope_cols, rope_pos_offset
    g.row_ssq = _p(row_ssq)
    g.ld_row_ssq = row_ssq.stride(-2) if row_ssq is not None else 0
    g.row_ssq_parts = row_norm_dim // 32 if row_ssq is not None else 0
    g.row_norm_dim = row_norm_dim
    t = AttnArgs()
    t.k, t.v, t.out = k, v, out
    t.k_row_stride, t.v_row_stride, t.k_batch_stride, t.v_batch_stride = kv_strides
    t.out_row_stride, t.out_batch_stride = out_strides
    t.B, t.H, t.Nq, t.Nk = B, H, rows_per_batch, Nk
    t.kv_len, t.q_len = _p(kv_len), _p(q_len)
    t.scale, t.softclamp, t.dtype = scale, softclamp, (BF16_SPLIT if split else BF16)
    t.out_split = 1 if split else 0
    _launch("qproj_xattn<%s>" % ("bf16x3" if split else "bf16"), (3.0 if split else 1.0) * (2.0 * M * (H * 65) * K + 4.0 * B * H * rows_per_batch * Nk * 64),
            M * K * 2 + H * 65 * K * 2 + M * H * 64 * 2 + 2 * B * Nk * H * 64 * 2,
            lambda: lib().v2a_qproj_xattn(C.byref(g), C.byref(t), stream_ptr()))


def linear_small(a, wt, bias, add, out, *, M, K, T, out_batch_stride, row_off, d, dup=0, regs=None, out_bf16=None):
    _launch("linear_small", 2.0 * M * K * d, M * (K + d) * 4,
            lambda: lib().v2a_linear_small(a.data_ptr(), M, K, wt.data_ptr(), _p(bias), _p(add), T, out.data_ptr(),
                                           out_batch_stride, row_off, d, dup, _p(regs), _p(out_bf16), stream_ptr()))


def fill_registers(out, regs, *, B, R, d, out_batch_stride):
    check(lib().v2a_fill_registers(out.data_ptr(), out_batch_stride, regs.data_ptr(), B, R, d, stream_ptr()))


def time_cond(t, fourier_w, wt, bias, out, *, S, d):
    check(lib().v2a_time_cond(t.data_ptr(), S, fourier_w.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                              out.data_ptr(), d, stream_ptr()))


def apg_reduce(pred, apg, *, B, T, C_, pred_batch_stride, row_off, valid_rows=None):
    """valid_rows: device int32[1] or None: the rows of every clip that enter the sums (a bucketed plan pads T behind them)."""
    check(lib().v2a_apg_reduce(pred.data_ptr(), apg.data_ptr(), B, T, C_, pred_batch_stride, row_off, _p(valid_rows), stream_ptr()))


def cfg_euler(y, pred, *, B, T, C_, pred_batch_stride, row_off, cfg_strength, dt, step=None, apg=None, keep=0.0):
    _launch("cfg_euler", 0.0, 16.0 * B * T * C_,
            lambda: lib().v2a_cfg_euler(y.data_ptr(), pred.data_ptr(), B, T, C_, pred_batch_stride, row_off,
                                        float(cfg_strength), dt.data_ptr(), _p(step), _p(apg), float(keep), stream_ptr()))


def step_advance(step):
    check(lib().v2a_step_advance(step.data_ptr(), stream_ptr()))


def split_bf16(x, y, *, rows, d, ldx=None, ldy=None):
    """fp32 (rows, d) -> bf16 (rows, 2*d) hi | lo planes."""
    _launch("split_bf16", 0.0, rows * d * 8, lambda: lib().v2a_split_bf16(x.data_ptr(), ldx or d, y.data_ptr(), ldy or 2 * d, rows, d, stream_ptr()))


def cast_bf16(x, y):
    _launch("cast_bf16", 0.0, x.numel() * 6, lambda: lib().v2a_cast_bf16(x.data_ptr(), y.data_ptr(), x.numel(), stream_ptr()))


# ---- N2: Video2Roll frame encoder ----------------------------------------------------------------
def im2col(x, col, *, B, H, W, C_, kh, kw, stride, pad, Ho, Wo, ldo, window_t=0, window_first=0):
    K = kh * kw * C_
    _launch("im2col<%s>" % ("f32" if col.dtype == torch.float32 else "bf16"), 0.0, B * Ho * Wo * (4.0 * K + col.element_size() * ldo),
            lambda: lib().v2a_im2col(x.data_ptr(), B, H, W, C_, kh, kw, stride, pad, Ho, Wo, col.data_ptr(), ldo,
                                     dt_code(col.dtype), window_t, window_first, stream_ptr()))


def frames_pack(frames, out, *, T, H, W, kw, stride, pad, Wo):
    _launch("frames_pack", 0.0, 4.0 * T * H * W + 32.0 * (T + 4) * Wo * (H + 2 * pad),
            lambda: lib().v2a_frames_pack(frames.data_ptr(), out.data_ptr(), T, H, W, kw, stride, pad, Wo, stream_ptr()))


def pool2d(x, out, *, B, H, W, C_, k, stride, pad, mode, Ho, Wo, out_bf16=None, in_border=0, out_border=0):
    _launch("pool2d", 0.0, 4.0 * B * C_ * (H * W + Ho * Wo),
            lambda: lib().v2a_pool2d(x.data_ptr(), out.data_ptr(), _p(out_bf16), B, H, W, C_, k, stride, pad, mode, Ho, Wo,
                                     in_border, out_border, stream_ptr()))


def roll_head(args: "RollHeadArgs"):
    _launch("roll_head", 0.0, 4.0 * args.B * args.P * (3 * 128 + 64), lambda: lib().v2a_roll_head(C.byref(args), stream_ptr()))


def roll_expand(roll, out, *, B, t, notes, rep, l):
    check(lib().v2a_roll_expand(roll.data_ptr(), out.data_ptr(), B, t, notes, rep, l, stream_ptr()))


# ---- N1: Encodec decoder (vocoder) ---------------------------------------------------------------
def elu_pad(x, out, *, T, C_, pad, reflect, act=True):
    _launch("elu_pad", 0.0, 4.0 * C_ * (2 * T + pad),
            lambda: lib().v2a_elu_pad(x.data_ptr(), out.data_ptr(), T, C_, pad, 1 if reflect else 0, 1 if act else 0, stream_ptr()))


def lstm_layer(gates_x, w_hh, h, workspace, *, T, H, resid=None, y=None):
    _launch("lstm_layer", 2.0 * T * 4 * H * H, 4.0 * T * 6 * H,
            lambda: lib().v2a_lstm_layer(gates_x.data_ptr(), w_hh.data_ptr(), h.data_ptr(), _p(resid), _p(y), T, H,
                                         workspace.data_ptr(), stream_ptr()))


def lstm2(gates_x0, w_hh0, w_ih1, bias1, w_hh1, y, workspace, *, T, H, resid=None):
    _launch("lstm2", 2.0 * T * 4 * H * H * 3, 4.0 * T * 6 * H,
            lambda: lib().v2a_lstm2(gates_x0.data_ptr(), w_hh0.data_ptr(), w_ih1.data_ptr(), bias1.data_ptr(), w_hh1.data_ptr(),
                                    _p(resid), y.data_ptr(), T, H, workspace.data_ptr(), stream_ptr()))
