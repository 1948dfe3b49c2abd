"""ctypes loader for libirbfdl.so (the C ABI of include/irb_fdl.h).

The library is built in-tree by `make -C irbaboon_amd/csrc` (see __graft_entry__.build).
There is no Python or CPU fallback: if the shared object is missing or a HIP device
is absent, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("IRB_LIB") or os.path.join(_HERE, "libirbfdl.so")     # IRB_LIB: another build of the same library (A/B measurements)

IRB_OK, IRB_ERR_ARG, IRB_ERR_HIP, IRB_ERR_LAYOUT, IRB_ERR_STATE = 0, -1, -2, -3, -4
FORCE_FUSED, NO_PAIRING, FORCE_COLS, HOST_IO, STEPWISE, NO_RT_KERNEL, RESIDENT = 0x1, 0x8, 0x10, 0x20, 0x40, 0x80, 0x100

fpp = C.POINTER(C.POINTER(C.c_float))
fp = C.POINTER(C.c_float)


class IrbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"irb error {code}: {msg}")
        self.code = code


class FdlConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("n_instances", C.c_int), ("channels", C.c_int), ("block", C.c_int),
                ("max_ir_len", C.c_int), ("max_blocks_per_call", C.c_int), ("stream", C.c_void_p),
                ("flags", C.c_uint)]


class FdlInfo(C.Structure):
    _fields_ = [("fft_len", C.c_int), ("partitions", C.c_int), ("ring_slots", C.c_int), ("streams", C.c_int),
                ("splits", C.c_int), ("device_bytes", C.c_size_t),
                ("algorithmic_bytes_per_block_step", C.c_double)]


class ConvolveProfile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("total_ms", "setup_ms", "register_ms", "pack_ms", "unpack_ms", "unpack_helper_ms", "wait_ms", "wait_helper_ms",
                                          "stall_ms", "fft_in_ms", "mac_ms", "fft_out_ms", "device_ms")] + \
               [(n, C.c_int) for n in ("direct", "chunks", "blocks_per_chunk", "pieces", "steps", "helper_thread", "engine_created",
                                       "mac_on_matrix_cores", "mac_launches")] + \
               [("pack_helper_ms", C.c_double), ("registered_ranges", C.c_int), ("threads", C.c_int), ("in_chunks", C.c_int)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/irb_fdl.h and include/irb_deconv.h declare: (name, restype, argtypes)
SYMBOLS = [
    ("irb_fdl_create", C.c_int, [C.POINTER(FdlConfig), C.POINTER(C.c_void_p)]),
    ("irb_fdl_destroy", None, [C.c_void_p]),
    ("irb_fdl_get_info", C.c_int, [C.c_void_p, C.POINTER(FdlInfo)]),
    ("irb_fdl_set_ir", C.c_int, [C.c_void_p, C.c_int, fpp, C.c_int, C.c_int]),
    ("irb_fdl_set_ir_partition", C.c_int, [C.c_void_p, C.c_int, C.c_int, fp]),
    ("irb_fdl_get_ir_spectra", C.c_int, [C.c_void_p, C.c_int, C.c_int, fp, C.c_int]),
    ("irb_fdl_set_ir_spectra", C.c_int, [C.c_void_p, C.c_int, fpp, C.c_int, C.c_int]),
    ("irb_fdl_reset", C.c_int, [C.c_void_p]),
    ("irb_fdl_process", C.c_int, [C.c_void_p, fpp, fpp, C.c_int]),
    ("irb_fdl_process_refresh", C.c_int, [C.c_void_p, C.c_int, fpp, fpp, C.c_int, fp, C.c_int]),
    ("irb_fdl_get_counters", C.c_int, [C.c_void_p] + [C.POINTER(C.c_ulonglong)] * 3),
    ("irb_fdl_process_packed", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]),
    ("irb_host_alloc", C.c_void_p, [C.c_size_t]),
    ("irb_host_free", None, [C.c_void_p]),
    ("irb_fdl_process_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]),
    ("irb_fdl_synchronize", C.c_int, [C.c_void_p]),
    ("irb_fdl_mark", C.c_int, [C.c_void_p, C.c_int]),
    ("irb_fdl_elapsed_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    ("irb_fdl_device_io", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_long)]),
    ("irb_fdl_upload", C.c_int, [C.c_void_p, C.c_void_p, fp, C.c_size_t]),
    ("irb_fdl_download", C.c_int, [C.c_void_p, fp, C.c_void_p, C.c_size_t]),
    ("irb_fdl_multi_create", C.c_int, [C.POINTER(FdlConfig), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    ("irb_fdl_multi_destroy", None, [C.c_void_p]),
    ("irb_fdl_multi_owner", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("irb_fdl_multi_set_ir", C.c_int, [C.c_void_p, C.c_int, fpp, C.c_int, C.c_int]),
    ("irb_fdl_multi_reset", C.c_int, [C.c_void_p]),
    ("irb_fdl_multi_process", C.c_int, [C.c_void_p, fpp, fpp, C.c_int]),
    ("irb_convolve_periodic", C.c_int, [C.c_int, fpp, C.c_int, C.c_int, fpp, C.c_int, C.c_int, C.c_int, C.c_int, fpp]),
    ("irb_device_quiet_begin", C.c_int, [C.c_int]),
    ("irb_device_quiet_end", C.c_int, [C.c_int]),
    ("irb_convolve_last_profile", C.c_int, [C.POINTER(ConvolveProfile)]),
    ("irb_convolve_profile_events", C.c_int, [C.c_int]),
    ("irb_fdl_mac_kernel_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("irb_rt_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("irb_rt_destroy", None, [C.c_void_p]),
    ("irb_rt_prepare_to_play", C.c_int, [C.c_void_p, C.c_int]),
    ("irb_rt_release_resources", C.c_int, [C.c_void_p]),
    ("irb_rt_set_ir", C.c_int, [C.c_void_p, fp, C.c_int]),
    ("irb_rt_process_block", C.c_int, [C.c_void_p, fpp, C.c_int]),
    ("irb_rt_process_block_bypassed", C.c_int, [C.c_void_p, fpp, C.c_int]),
    ("irb_rt_time_callbacks", C.c_int, [C.c_void_p, C.c_int, C.c_int, fp]),
    ("irb_rt_counters", C.c_int, [C.c_void_p] + [C.POINTER(C.c_ulonglong)] * 4),
    ("irb_rt_latency", C.c_int, [C.c_void_p]),
    ("irb_rt_partitions", C.c_int, [C.c_void_p]),
    # include/irb_deconv.h
    ("irb_next_power_of_two", C.c_int, [C.c_int]),
    ("irb_fft_transform", C.c_int, [C.c_int, fp, C.c_int, fp]),
    ("irb_fft_transform_ampl_phase", C.c_int, [C.c_int, fp, C.c_int, fp]),
    ("irb_fft_inv_transform", C.c_int, [C.c_int, fp, C.c_int, fp]),
    ("irb_averaging_filter", None, [fp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
    ("irb_shifteroo", None, [fp, C.c_int]),
    ("irb_deconvolve", C.c_int, [C.c_int, fp, C.c_int, fp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, fp, C.c_int]),
    ("irb_invert_filter", C.c_int, [C.c_int, fp, C.c_int, C.c_int, fp, C.c_int]),
    ("irb_convolve_nonperiodic", C.c_int, [C.c_int, fpp, C.c_int, C.c_int, fpp, C.c_int, C.c_int, fpp]),
    ("irb_ir_to_real_fft_raw", C.c_int, [C.c_int, fp, C.c_int, C.c_int, fp, C.c_int, C.POINTER(C.c_int)]),
    ("irb_sweep_generate", C.c_int, [C.c_double] * 5 + [C.POINTER(C.c_double), C.c_int]),
    ("irb_sweep_generate_inv", C.c_int, [C.c_double] * 5 + [C.POINTER(C.c_double), C.c_int]),
    ("irb_sweep_sample_index_at_freq", C.c_int, [C.c_double] * 5),
    ("irb_sweep_freq_at_sample_index", C.c_double, [C.c_int] + [C.c_double] * 4),
    ("irb_sweep_lin_fadeout", C.c_int, [C.POINTER(C.c_double), C.c_int, C.c_int]),
    ("irb_release_cached_plans", None, []),
    ("irb_lfft_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("irb_lfft_destroy", None, [C.c_void_p]),
    ("irb_lfft_buffers", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("irb_lfft_invalidate_spec", C.c_int, [C.c_void_p]),
    ("irb_lfft_forward_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("irb_lfft_inverse_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("irb_lfft_synchronize", C.c_int, [C.c_void_p]),
    ("irb_lfft_mark", C.c_int, [C.c_void_p, C.c_int]),
    ("irb_lfft_elapsed_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    ("irb_last_error", C.c_char_p, []),
    ("irb_version", C.c_char_p, []),
]

_lib = None


def lib():
    """Load libirbfdl.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            # not a fallback: the only implementation is the HIP library, so build it from source
            import subprocess
            r = subprocess.run(["make", "-s", "-j3", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(SO_PATH):
                raise ImportError(f"{SO_PATH} is missing and `make -C irbaboon_amd/csrc` failed "
                                  f"(there is no CPU fallback):\n{r.stderr[-2000:]}")
        l = C.CDLL(SO_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != IRB_OK:
        raise IrbError(rc, lib().irb_last_error().decode("utf-8", "replace"))


def planar_ptrs(a):
    """float32 C-contiguous [rows][n] -> float** over the rows (the array must outlive the call)."""
    return (fp * a.shape[0])(*[a[i].ctypes.data_as(fp) for i in range(a.shape[0])])
