"""ctypes binding of libninpol_amd.so (the C ABI in include/ninpol_amd.h).

There is no fallback: if the shared object is missing or does not load, importing the product
path fails loudly.  (`python -m ninpol_amd.build` / `__graft_entry__.build()` produce it.)
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NINPOL_AMD_LIB: another build of the same library (A/B timing of kernel variants in one GPU session)
LIB_PATH = os.environ.get("NINPOL_AMD_LIB") or os.path.join(HERE, "libninpol_amd.so")

NIN_OK = 0
NIN_EINVAL, NIN_ENOMEM, NIN_EHIP, NIN_ENODEVICE, NIN_ERANGE, NIN_ESTATE = -1, -2, -3, -4, -5, -6
METHOD_ID = {"gls": 0, "idw": 1, "ls": 2}

EXPORTS = (
    "nin_last_error", "nin_version", "nin_grid_create", "nin_grid_create_on_device", "nin_grid_destroy", "nin_grid_scalar",
    "nin_grid_array_info", "nin_grid_array_copy", "nin_device_count", "nin_grid_to_device", "nin_grid_device",
    "nin_fields_set", "nin_weights_device", "nin_weights_host", "nin_csr_compact_host", "nin_interpolate_csr_host", "nin_apply_host",
    "nin_apply_device", "nin_apply_fields_host", "nin_pack_connectivity", "nin_pack_table_row", "nin_diff_mag",
    "nin_algorithmic_bytes", "nin_kernel_name", "nin_gls_plan", "nin_gls_plan_flops", "nin_host_alloc", "nin_host_free", "nin_hash64",
    "nin_grid_release_scratch",
    "nin_exchange_create", "nin_exchange_destroy", "nin_exchange_handle", "nin_exchange_connect", "nin_exchange_push",
    "nin_exchange_wait_sent", "nin_exchange_buffer", "nin_exchange_slot_bytes",
)

_lib = None


class NinpolError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libninpol_amd error {code}: {text}")
        self.code = code
        self.text = text


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m ninpol_amd.build` "
                          "(there is no CPU fallback for the weight kernels)")
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, cp, dp = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p
    L.nin_last_error.restype = cp
    L.nin_version.restype = cp
    L.nin_grid_create.argtypes = [i64, i64, i64] + [vp] * 9 + [i32, i32, i32, ctypes.POINTER(vp)]
    L.nin_grid_create_on_device.argtypes = [i64, i64, i64] + [vp] * 9 + [i32, i32, i32, ctypes.POINTER(vp)]
    L.nin_grid_destroy.argtypes = [vp]
    L.nin_grid_destroy.restype = None
    L.nin_grid_scalar.argtypes = [vp, cp]
    L.nin_grid_scalar.restype = i64
    L.nin_grid_array_info.argtypes = [vp, cp, ctypes.POINTER(i64), ctypes.POINTER(i32)]
    L.nin_grid_array_copy.argtypes = [vp, cp, vp, i64]
    L.nin_device_count.argtypes = [ctypes.POINTER(i32)]
    L.nin_grid_to_device.argtypes = [vp, i32]
    L.nin_grid_device.argtypes = [vp]
    L.nin_fields_set.argtypes = [vp, dp, dp, dp, dp]
    L.nin_weights_device.argtypes = [vp, i32, vp, i64, i32, vp, vp, vp]
    L.nin_weights_host.argtypes = [vp, i32, vp, i64, i32, vp, vp]
    L.nin_csr_compact_host.argtypes = [vp, vp, vp, vp, vp, ctypes.POINTER(i64), vp]
    L.nin_interpolate_csr_host.argtypes = [vp, i32, vp, vp, vp, ctypes.POINTER(i64), vp]
    L.nin_apply_host.argtypes = [vp, i32, vp, vp, vp]
    L.nin_apply_device.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    L.nin_apply_fields_host.argtypes = [vp, i32, vp, i32, vp, vp]
    L.nin_pack_connectivity.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    L.nin_pack_table_row.argtypes = [vp, i64, i64, i64, vp]
    L.nin_diff_mag.argtypes = [vp, i64, vp]
    L.nin_algorithmic_bytes.argtypes = [vp, i32]
    L.nin_algorithmic_bytes.restype = i64
    L.nin_kernel_name.argtypes = [i32]
    L.nin_kernel_name.restype = cp
    L.nin_gls_plan.argtypes = [vp, vp]
    L.nin_gls_plan_flops.argtypes = [vp, vp, vp, vp]
    L.nin_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    L.nin_host_free.argtypes = [vp]
    L.nin_hash64.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    L.nin_grid_release_scratch.argtypes = [vp]
    L.nin_exchange_create.argtypes = [i32, i32, i32, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.nin_exchange_destroy.argtypes = [vp]
    L.nin_exchange_destroy.restype = None
    L.nin_exchange_handle.argtypes = [vp, vp]
    L.nin_exchange_connect.argtypes = [vp, vp]
    L.nin_exchange_push.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp]
    L.nin_exchange_wait_sent.argtypes = [vp, vp, i32]
    L.nin_exchange_buffer.argtypes = [vp]
    L.nin_exchange_buffer.restype = vp
    L.nin_exchange_slot_bytes.argtypes = [vp]
    L.nin_exchange_slot_bytes.restype = ctypes.c_size_t
    _lib = L
    return L


def check(rc):
    if rc != NIN_OK:
        raise NinpolError(rc, load().nin_last_error().decode())


def device_count():
    n = ctypes.c_int(0)
    rc = load().nin_device_count(ctypes.byref(n))
    return n.value if rc == NIN_OK else 0


class PinnedPool:
    """Page-locked host buffers (nin_host_alloc) handed out as numpy arrays and taken back when the last view of an
    array dies: device-to-host copies into them run at PCIe rate, and pinning a gigabyte costs ~50 ms, so buffers are
    kept (up to `keep_bytes`) for the next call.  Not thread-safe (neither is the Interpolator)."""

    def __init__(self, keep_bytes=None):
        """keep_bytes: how much page-locked memory may sit idle in the pool (default 4 GiB, or the environment's
        NINPOL_AMD_PINNED_KEEP_BYTES); settable at any time, `trim()` gives idle buffers back to the system."""
        self.free = []          # (bytes, address)
        if keep_bytes is None:
            keep_bytes = int(os.environ.get("NINPOL_AMD_PINNED_KEEP_BYTES", 4 << 30))
        self.keep_bytes = int(keep_bytes)

    def idle_bytes(self):
        return sum(b for b, _ in self.free)

    def trim(self, to_bytes=0):
        """Free idle page-locked buffers, largest first, until at most `to_bytes` stay.  (Buffers still referenced by
        arrays handed out -- e.g. the CSR a caller keeps -- are not touched; they come back here when the arrays die.)"""
        self.free.sort()
        while self.free and self.idle_bytes() > to_bytes:
            _, addr = self.free.pop()
            load().nin_host_free(addr)

    def empty(self, n, dtype):
        import weakref
        import numpy as np
        nbytes = max(int(n) * np.dtype(dtype).itemsize, 1)
        best = None
        for i, (b, _) in enumerate(self.free):
            if nbytes <= b <= 2 * nbytes + 4096 and (best is None or b < self.free[best][0]):
                best = i
        if best is not None:
            size, addr = self.free.pop(best)
        else:
            p = ctypes.c_void_p()
            if load().nin_host_alloc(nbytes, ctypes.byref(p)) != 0 or not p.value:
                return np.empty(int(n), dtype=dtype)      # no page-locked memory to be had: an ordinary array (slower copies)
            size, addr = nbytes, p.value
        owner = (ctypes.c_char * size).from_address(addr)
        weakref.finalize(owner, self._release, size, addr)
        return np.frombuffer(owner, dtype=dtype, count=int(n))

    def _release(self, size, addr):
        try:
            if sum(b for b, _ in self.free) + size <= self.keep_bytes:
                self.free.append((size, addr))
            else:
                load().nin_host_free(addr)
        except Exception:       # interpreter shutdown
            pass
