"""ctypes view of kmers_amd/lib/libkmers.so: the reference's six symbols plus the extended ABI of include/kmers_hip.h."""
import ctypes as C
import os
from ctypes.util import find_library

_PKG = os.path.dirname(os.path.abspath(__file__))
_handle = None

X_ARGS = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64),
          C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
Y_ARGS = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]

ERRORS = {
    -1: "Failed to open FASTA file",
    -2: "Failed to parse FASTA file: insufficient max_nz",
    -10: "libkmers: no usable HIP device (this library has no CPU fallback)",
    -11: "libkmers: HIP runtime error (see stderr)",
    -12: "libkmers: bad argument",
    -13: "libkmers: input too large for one device call (more than 2^32-65536 positions)",
    -14: "libkmers: workspace too small",
    -15: "libkmers: host allocation failed",
}


def lib_path():
    """The in-tree build first (kmers_amd/lib/libkmers.so), then KMERS_LIB, then the system search path."""
    cand = [os.environ.get("KMERS_LIB"), os.path.join(_PKG, "lib", "libkmers.so")]
    for c in cand:
        if c and os.path.exists(c):
            return c
    return find_library("kmers")


def _share_hip_runtime_with_torch():
    """torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; a process that initialises two copies of
    the HIP runtime finds no GPU in the second one.  Importing torch BEFORE libkmers.so is loaded makes the dynamic
    linker resolve libkmers' NEEDED libamdhip64.so.7 to the copy torch already mapped, so bench.py and the
    device-resident API (which use torch tensors as HBM allocations) share one runtime with the library.
    A process that never uses torch can set KMERS_NO_TORCH=1 and run on the system ROCm runtime alone."""
    if os.environ.get("KMERS_NO_TORCH"):
        return
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def handle():
    global _handle
    if _handle is not None:
        return _handle
    _share_hip_runtime_with_torch()
    path = lib_path()
    if not path:
        raise ImportError("Unable to find 'libkmers' shared object: build it with `python -m kmers_amd.build`")
    lib = C.CDLL(path)
    for name in ("fasta_to_kmers_csr", "fasta_to_kmers_csr_cat_subseq"):
        f = getattr(lib, name)
        f.restype, f.argtypes = C.c_int, X_ARGS
    for name in ("fasta_count_kmers", "fastq_gz_count_kmers"):
        f = getattr(lib, name)
        f.restype, f.argtypes = C.c_int, Y_ARGS
    lib.kmer_to_id.restype, lib.kmer_to_id.argtypes = C.c_uint32, [C.c_char_p, C.c_int]
    lib.valid_kmer.restype, lib.valid_kmer.argtypes = C.c_bool, [C.c_char_p, C.c_int]

    vp = C.c_void_p
    lib.kmers_store_ingest.restype = C.c_int
    lib.kmers_store_ingest.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.kmers_store_ingest_bytes.restype = C.c_int
    lib.kmers_store_ingest_bytes.argtypes = lib.kmers_store_ingest.argtypes
    lib.kmers_store_set_numbering.restype, lib.kmers_store_set_numbering.argtypes = None, [vp, C.c_uint64, C.c_uint64]
    lib.kmers_store_free.restype, lib.kmers_store_free.argtypes = None, [vp]
    for name in ("n_pos", "n_rec", "n_rec_total", "first_rec", "n_bases"):
        f = getattr(lib, "kmers_store_" + name)
        f.restype, f.argtypes = C.c_uint64, [vp]
    lib.kmers_store_n_windows.restype, lib.kmers_store_n_windows.argtypes = C.c_uint64, [vp, C.c_int]
    for name in ("bases", "valid", "rec_start"):
        f = getattr(lib, "kmers_store_" + name)
        f.restype, f.argtypes = vp, [vp]
    for name in ("bases_words", "valid_words"):
        f = getattr(lib, "kmers_store_" + name)
        f.restype, f.argtypes = C.c_size_t, [vp]
    lib.kmers_hip_store_upload.restype, lib.kmers_hip_store_upload.argtypes = C.c_int, [vp, vp, vp, vp, vp]
    lib.kmers_hip_device_count.restype, lib.kmers_hip_device_count.argtypes = C.c_int, []
    lib.kmers_hip_count_y.restype = C.c_int
    lib.kmers_hip_count_y.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_uint64, C.c_int, vp, vp]
    lib.kmers_hip_y_workspace_bytes.restype, lib.kmers_hip_y_workspace_bytes.argtypes = C.c_size_t, [C.c_uint64, C.c_int]
    lib.kmers_hip_count_y_ws.restype = C.c_int
    lib.kmers_hip_count_y_ws.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_uint64, C.c_int, vp, vp, C.c_size_t, vp]
    lib.kmers_hip_count_y_static.restype = C.c_int
    lib.kmers_hip_count_y_static.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_uint64, C.c_int, vp, vp]
    lib.kmers_hip_x_workspace_bytes.restype = C.c_size_t
    lib.kmers_hip_x_workspace_bytes.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    lib.kmers_hip_build_x.restype = C.c_int
    lib.kmers_hip_build_x.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                      vp, vp, vp, C.c_uint64, vp, C.c_size_t, C.POINTER(C.c_uint64), vp]
    lib.kmers_csr_merge.restype = C.c_int
    lib.kmers_csr_merge.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp, vp]
    lib.kmers_csr_place.restype = C.c_int
    lib.kmers_csr_place.argtypes = [C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.kmers_csr_place_rows.restype = C.c_int
    lib.kmers_csr_place_rows.argtypes = [C.c_uint64, vp, vp, vp, vp, vp, vp, vp]
    lib.kmers_npz_write_csr.restype = C.c_int
    lib.kmers_npz_write_csr.argtypes = [C.c_char_p, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.kmers_npz_write_csr_i32.restype = C.c_int
    lib.kmers_npz_write_csr_i32.argtypes = [C.c_char_p, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.kmers_fasta_to_kmers_csr_i32.restype = C.c_int
    lib.kmers_fasta_to_kmers_csr_i32.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_uint64,
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    lib.kmers_hip_model_create.restype = C.c_int
    lib.kmers_hip_model_create.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, C.POINTER(vp)]
    lib.kmers_hip_simulate_reads.restype = C.c_int
    lib.kmers_hip_simulate_reads.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_uint32, vp, C.c_size_t, vp,
                                             C.c_size_t, vp]
    lib.kmers_sim_starts.restype = C.c_int
    lib.kmers_sim_starts.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.kmers_hip_model_create_shard.restype = C.c_int
    lib.kmers_hip_model_create_shard.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, C.POINTER(vp)]
    lib.kmers_hip_model_xtheta_into.restype = C.c_int
    lib.kmers_hip_model_xtheta_into.argtypes = [vp, vp, vp, vp]
    lib.kmers_hip_model_grad_from.restype = C.c_int
    lib.kmers_hip_model_grad_from.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_double), vp, vp]
    lib.kmers_hip_model_free.restype, lib.kmers_hip_model_free.argtypes = None, [vp]
    for name in ("n_masked", "nnz"):
        f = getattr(lib, "kmers_hip_model_" + name)
        f.restype, f.argtypes = C.c_uint64, [vp]
    lib.kmers_hip_model_sum_y.restype, lib.kmers_hip_model_sum_y.argtypes = C.c_int64, [vp]
    for name in ("rows", "xtheta"):
        f = getattr(lib, "kmers_hip_model_" + name)
        f.restype, f.argtypes = vp, [vp]
    lib.kmers_hip_model_read.restype, lib.kmers_hip_model_read.argtypes = C.c_int, [vp, C.c_int, vp]
    lib.kmers_hip_model_logp_grad.restype = C.c_int
    lib.kmers_hip_model_logp_grad.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_double), vp, vp]
    lib.kmers_x_range_count.restype, lib.kmers_x_range_count.argtypes = C.c_uint64, [C.c_uint64] * 4
    lib.kmers_last_call_stages.restype, lib.kmers_last_call_stages.argtypes = C.c_int, [C.c_char_p, C.c_size_t]
    lib.kmers_hip_debug_knob.restype, lib.kmers_hip_debug_knob.argtypes = C.c_int, [C.c_char_p, C.c_longlong]
    lib.kmers_hip_profile_enable.restype, lib.kmers_hip_profile_enable.argtypes = None, [C.c_int]
    lib.kmers_hip_profile_count.restype, lib.kmers_hip_profile_count.argtypes = C.c_int, []
    lib.kmers_hip_profile_get.restype = C.c_int
    lib.kmers_hip_profile_get.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    _handle = lib
    return lib


def profile(enable):
    handle().kmers_hip_profile_enable(1 if enable else 0)


def profile_table():
    """[(kernel name, total ms, launches)] accumulated since profile(True)."""
    lib = handle()
    out = []
    for i in range(lib.kmers_hip_profile_count()):
        name = C.create_string_buffer(64)
        ms, calls = C.c_double(0), C.c_uint64(0)
        if lib.kmers_hip_profile_get(i, name, 64, C.byref(ms), C.byref(calls)) == 0:
            out.append((name.value.decode(), ms.value, int(calls.value)))
    return out


def last_call_stages():
    """{stage: ms} of the most recent file-level call (kmers_last_call_stages)."""
    buf = C.create_string_buffer(1024)
    handle().kmers_last_call_stages(buf, 1024)
    out = {}
    for item in buf.value.decode().split(";"):
        if "=" in item:
            k, v = item.rsplit("=", 1)
            out[k] = out.get(k, 0.0) + float(v)
    return out


def check(rc, what="libkmers"):
    if rc == 0:
        return
    raise RuntimeError(ERRORS.get(rc, f"{what}: error {rc}"))


def cfiles(paths):
    arr = (C.c_char_p * len(paths))()
    arr[:] = [p.encode("UTF-8") for p in paths]
    return arr
