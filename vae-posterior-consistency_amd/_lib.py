"""ctypes binding of libvpc_hip.so (C ABI declared in include/vpc.h).

The product path has NO CPU fallback: if the HIP library is missing or a launch fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import lru_cache

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VPC_LIB") or os.path.join(_HERE, "csrc", "libvpc_hip.so")  # VPC_LIB: diagnostic builds only

_ERR = {1: "bad argument (null / misaligned pointer or bad count)", 2: "unsupported shape (d > 128 or L > 15)",
        3: "HIP runtime error"}


class VpcError(RuntimeError):
    pass


P = C.c_void_p
PP = C.POINTER(C.c_void_p)
I = C.c_int
L_ = C.c_long
F = C.c_float
IP = C.POINTER(C.c_int)
ULL = C.c_ulonglong

# name -> argtypes, in the order of include/vpc.h
_PROTOS = {
    "vpc_layout_sizes": [I, I, I, IP, IP, IP, IP, IP, IP, IP, IP],
    "vpc_build_indices": [I, I, I, P, P, P],
    "vpc_num_cus": [],
    "vpc_max_partial_blocks": [],
    "vpc_pack_weights": [P, P, P, I, P],
    "vpc_reduce_partials": [P, I, L_, P, P, I, F, P],
    "vpc_adam_step": [P, P, P, P, I, F, F, F, F, L_, P, P, P, P, P, P],
    "vpc_layout_sizes_bf16": [I, I, I, IP, IP],
    "vpc_build_indices_bf16": [I, I, I, P, P],
    "vpc_pack_weights_bf16": [P, P, P, I, P],
    "vpc_step_small_max_rows": [],
    "vpc_step_small_f32": [P, P, P, I, PP, PP, C.POINTER(F), C.POINTER(F), PP, P, F, F, F, F, F, F, P, P, P, IP, L_, I, I, P],
    "vpc_step_small_draw_f32": [P, P, P, I, PP, PP, C.POINTER(F), C.POINTER(F), PP, P, F, F, F, F, F, F, P, P, P, IP, L_, I, I,
                                P, F, P, L_, ULL, ULL, ULL, P, L_, L_, L_, L_, I, P],
    "vpc_step_fused_applicable": [L_, I, I, I],
    "vpc_step_layout_bf16": [I, I, IP, IP],
    "vpc_step_build_indices_bf16": [I, I, P, P],
    "vpc_step_pack_weights_bf16": [P, P, P, I, P],
    "vpc_step_workspace_floats": [L_],
    "vpc_step_fused_bf16": [P, P, I, PP, PP, C.POINTER(F), C.POINTER(F), PP, P, F, F, F, F, F, F, P, P, P, P, IP, L_, I, I, P],
    "vpc_encoder_fwd": [P, P, I, PP, PP, PP, PP, PP, PP, PP, I, I, I, L_, I, I, P],
    "vpc_encoder_bwd": [P, P, I, PP, PP, PP, PP, PP, I, I, I, P, IP, L_, I, I, P],
    "vpc_decoder_fwd": [P, P, P, L_, I, I, P],
    "vpc_decoder_bwd": [P, P, P, P, P, IP, L_, I, I, P],
    "vpc_loss_fwd_bwd": [P, I, PP, PP, PP, C.POINTER(F), C.POINTER(F), PP, PP, P, F, F, F, F, F, F, PP, PP, PP, P, I,
                         IP, L_, I, I, P],
    "vpc_decoder_fused": [P, P, I, PP, PP, C.POINTER(F), C.POINTER(F), PP, PP, PP, P, F, F, F, F, F, F, PP, PP, I, I,
                          P, P, IP, L_, I, I, P],
    "vpc_loss_finalize": [P, I, F, F, F, F, F, F, F, L_, L_, I, P, P, P],
    "vpc_build_inverse_maps": [P, I, I, L_, L_, P, P],
    "vpc_reduce_step": [P, I, L_, P, I, L_, P, P, P, I, I, P, I, F, F, F, F, F, F, F, L_, L_, I, P, P, P, C.c_longlong,
                        P],
    "vpc_reduce_step_adam": [P, I, L_, P, I, L_, P, P, P, I, I, P, I, F, F, F, F, F, F, F, L_, L_, I, P, P, P, P, P, F, F,
                             F, F, L_, P, P, P],
    "vpc_reduce_step_adam_bf16c": [P, I, L_, P, I, L_, P, P, P, I, I, P, I, F, F, F, F, F, F, F, L_, L_, I, P, P, P, P, P, F, F,
                             F, F, L_, P, P, P],
    "vpc_draw_mask": [P, P, L_, F, ULL, ULL, L_, P],
    "vpc_draw_step": [P, P, L_, F, P, L_, ULL, ULL, ULL, P, L_, L_, L_, L_, I, P],
    "vpc_fill_normal": [P, L_, ULL, ULL, P, L_, L_, L_, I, P],
    "vpc_rccl_unique_id": [P],
    "vpc_rccl_comm_init": [P, I, I, PP],
    "vpc_allreduce_flat": [P, P, L_, P],
    "vpc_rccl_comm_destroy": [P],
    "vpc_reward_scratch": [I, I, I, C.POINTER(L_), C.POINTER(L_), C.POINTER(L_)],
    "vpc_reward_matrix": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    # MNAR path (config 3)
    "vpc_linear_fwd": [P, L_, P, P, P, L_, L_, I, I, I, I, I, P],
    "vpc_linear_dgrad": [P, L_, P, L_, I, I, P, P, L_, I, P, L_, L_, I, I, I, P],
    "vpc_linear_wgrad_scratch": [L_, I, I],
    "vpc_linear_wgrad": [P, L_, P, L_, I, I, P, L_, P, P, P, L_, L_, I, I, I, I, P],
    "vpc_linear_wgrad_reduce": [I, PP, C.POINTER(L_), IP, IP, PP, PP, IP, P],
    "vpc_nm_sample": [P, L_, P, P, L_, L_, I, I, P],
    "vpc_nm_sample_bwd": [P, L_, P, P, L_, P, L_, P, L_, L_, I, I, P],
    "vpc_nm_mul": [P, P, P, L_, P],
    "vpc_nm_loss_blocks": [L_],
    "vpc_nm_loss_scratch": [L_, I],
    "vpc_nm_loss": [P, P, P, P, P, L_, P, P, L_, P, P, L_, P, P, P, P, P, L_, P, P, L_, P, P, L_, P, P, I, P, P, L_, P,
                    P, P, P, C.c_longlong, I, L_, L_, I, I, I, C.c_double, P],
    "vpc_nm_prep": [P, P, P, P, L_, I, F, P, L_, ULL, ULL, ULL, P, L_, L_, L_, L_, I, P],
    "vpc_nmdec_applicable": [L_, I, I, I],
    "vpc_nmdec_layout": [L_, I, I, I, IP, C.POINTER(C.c_long), IP],
    "vpc_nmdec_build_indices": [I, I, I, P, P, I],
    "vpc_nmenc_fwd": [P, P, P, P, P, L_, I, I, P],
    "vpc_nmenc_bwd": [P, P, P, P, P, P, L_, P, P, L_, I, I, P],
    "vpc_nmenc_build_indices": [I, I, I, P, C.POINTER(C.c_long), I],
    "vpc_nm_fused_bwd_step": [P, P, P, P, P, P, P, P, L_, P, P, P, P, P, L_, P, P, P, I, P, P, P, L_, L_, I, I, I, C.c_double,
                              P, P, P, F, F, F, F, L_, P, P],
    "vpc_nmdec_step": [P, P, P, P, P, L_, P, P, P, P, P, P, P, I, P, P, P, P, C.c_longlong, L_, L_, I, I, I, C.c_double, P],
    # PNP / EDDI encoder front-end
    "vpc_eddi_fold": [P, P, P, P, P, I, I, P],
    "vpc_eddi_front_fwd": [P, P, P, P, P, L_, I, I, P],
    "vpc_eddi_front_scratch": [L_, I, I],
    "vpc_eddi_front_bwd": [P, P, P, P, P, P, P, P, P, L_, P, P, P, P, I, L_, I, I, P],
}
_RESTYPE_LONG = {"vpc_step_small_max_rows", "vpc_step_workspace_floats", "vpc_linear_wgrad_scratch", "vpc_nm_loss_scratch", "vpc_eddi_front_scratch"}

_lib = None


def lib():
    """Load the shared library (once).  Raises VpcError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VpcError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in _PROTOS.items():
            fn = getattr(handle, name)  # AttributeError here = header / library mismatch
            fn.argtypes = argtypes
            fn.restype = L_ if name in _RESTYPE_LONG else I
        _lib = handle
    return _lib


def exported_symbols():
    return list(_PROTOS)


def check(code: int, what: str):
    if code != 0:
        raise VpcError(f"{what} failed: {_ERR.get(code, code)}")


def ptr(t):
    """Device / host pointer of a tensor (or None) as c_void_p."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """HOST array of pointers for the per-pass arguments."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def farray(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """torch's CURRENT stream on the current device as a hipStream_t (queried on every launch: graph capture and user
    streams change it; the raw C query is ~10x cheaper than building a torch.cuda.Stream object per launch)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VpcError("this path runs only on the GPU (HIP kernels, no CPU fallback): got a CPU tensor; "
                           "move the model and its inputs to 'cuda'")


class Layout:
    """Sizes and index tables of the packed layouts for one (d, L)."""

    def __init__(self, d: int, L: int, mask_augm: bool = False):
        l = lib()
        self.mask_augm = int(bool(mask_augm))
        vals = [C.c_int() for _ in range(8)]
        check(l.vpc_layout_sizes(d, L, self.mask_augm, *[C.byref(v) for v in vals]), "vpc_layout_sizes")
        (self.enc_img, self.dec_img, self.n_enc, self.n_params, self.enc_part, self.dec_part, self.loss_terms,
         self.tile_rows) = [v.value for v in vals]
        self.d, self.L = d, L
        self.pack_idx = np.empty(self.n_params, np.int32)
        self.grad_idx = np.empty(self.n_params, np.int32)
        self.img_template = np.empty(self.enc_img + self.dec_img, np.float32)
        check(l.vpc_build_indices(d, L, self.mask_augm, self.pack_idx.ctypes.data_as(P), self.grad_idx.ctypes.data_as(P),
                                  self.img_template.ctypes.data_as(P)), "vpc_build_indices")
        self._dev = {}

    def device_tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.pack_idx).to(device), torch.from_numpy(self.grad_idx).to(device))
        return self._dev[key]

    def bf16_tables(self, device):
        """(pack_idx_bf [n_params] int32 on `device`, img_template_bf numpy [enc + dec floats], enc_img floats) of the
        bf16 weight images (precision 1 / 2: csrc/vpc_bf16.h)."""
        key = ("bf16", str(device))
        if key not in self._dev:
            l = lib()
            e, dd = C.c_int(), C.c_int()
            check(l.vpc_layout_sizes_bf16(self.d, self.L, self.mask_augm, C.byref(e), C.byref(dd)), "vpc_layout_sizes_bf16")
            idx = np.empty(self.n_params, np.int32)
            tmpl = np.empty(e.value + dd.value, np.float32)
            check(l.vpc_build_indices_bf16(self.d, self.L, self.mask_augm, idx.ctypes.data_as(P), tmpl.ctypes.data_as(P)),
                  "vpc_build_indices_bf16")
            self._dev[key] = (torch.from_numpy(idx).to(device), tmpl, e.value)
        return self._dev[key]

    def step_tables(self, device):
        """(pack_idx_c [n_params] int32 on `device`, img_template_c numpy [floats]) of the compact bf16 image of the
        whole-step kernel (csrc/vpc_step.hip)."""
        key = ("step", str(device))
        if key not in self._dev:
            l = lib()
            n = C.c_int()
            check(l.vpc_step_layout_bf16(self.d, self.L, C.byref(n), None), "vpc_step_layout_bf16")
            idx = np.empty(self.n_params, np.int32)
            tmpl = np.empty(n.value, np.float32)
            check(l.vpc_step_build_indices_bf16(self.d, self.L, idx.ctypes.data_as(P), tmpl.ctypes.data_as(P)),
                  "vpc_step_build_indices_bf16")
            self._dev[key] = (torch.from_numpy(idx).to(device), tmpl)
        return self._dev[key]

    def inverse_maps(self, device):
        """Caller-owned inverse maps (partial-block position -> flat parameter, -1 = padding) for the layout-order
        gradient reduction of vpc_reduce_step: [enc_part | dec_part] int32 on `device`, built once per device by the
        explicit entry point vpc_build_inverse_maps (the library itself allocates nothing)."""
        key = ("inv", str(device))
        if key not in self._dev:
            _, gidx = self.device_tables(device)
            inv = torch.empty(self.enc_part + self.dec_part, dtype=torch.int32, device=device)
            check(lib().vpc_build_inverse_maps(ptr(gidx), self.n_enc, self.n_params, self.enc_part, self.dec_part,
                                               ptr(inv), stream_ptr()), "vpc_build_inverse_maps")
            self._dev[key] = inv
        return self._dev[key]

    def nblocks(self, B: int, ncu: int) -> int:
        return min((B + self.tile_rows - 1) // self.tile_rows, ncu)


@lru_cache(maxsize=None)
def layout(d: int, L: int, mask_augm: bool = False) -> Layout:
    return Layout(d, L, mask_augm)


_NCU = None


def num_cus() -> int:
    global _NCU
    if _NCU is None:
        _NCU = int(lib().vpc_num_cus())
    return _NCU


def max_blocks() -> int:
    """Upper bound of the partial blocks any kernel writes (sizes the partial / loss-partial buffers)."""
    return int(lib().vpc_max_partial_blocks())


# Positions of the hidden units (index H = the constant-1 unit of the bias chain) inside the padded 112- / 64-wide
# workspaces and weight images (csrc/vpc_layout.h pos1 / pos2): the last, partly filled tile is laid out j-major so
# that the MFMA k-steps holding only padding can be skipped.  Only tests and debugging tools need this.
HIDDEN_POS1 = list(range(96)) + [96, 100, 104, 108, 97]
HIDDEN_POS2 = list(range(48)) + [48, 52, 56]
