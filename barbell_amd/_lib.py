"""Loader of the in-tree HIP library barbell_amd/libbarbell_amd.so (built by __graft_entry__.build()
or barbell_amd/csrc/build.sh).  There is no fallback: if the library is missing, importing the
compute entry points fails loudly."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BARBELL_AMD_SO") or os.path.join(_HERE, "libbarbell_amd.so")  # BARBELL_AMD_SO: an experimental build (dev knob)

# every symbol include/barbell_amd.h and include/barbell_amd_synth.h declare
EXPORTS = [
    "bb_create", "bb_create_policy", "bb_get_policy", "bb_destroy", "bb_n_groups", "bb_group_get_info", "bb_group_get_flank", "bb_group_get_pattern",
    "bb_annotate_batch", "bb_annotate_batch_dev", "bb_counts_len", "bb_counts", "bb_counts_dev", "bb_counts_reset",
    "bb_last_scan_stats", "bb_filter_twin", "bb_last_length_stats", "bb_pack_bases", "bb_annotate_batch_packed", "bb_last_host_syncs", "bb_host_phases", "bb_last_barcode_stats", "bb_n_kernels", "bb_kernel_name", "bb_last_kernel_ms", "bb_set_timing", "bb_last_dominant_kernel", "bb_strerror", "bb_last_error", "bb_build_trace_classes",
    "bb_synth_offsets", "bb_synth_reads_host", "bb_synth_reads_dev",
    "bb_filter_set", "bb_filter_rows", "bb_filter_rows_dev",
    "bb_inspect_rows", "bb_inspect_rows_dev",
    "bb_dev_malloc", "bb_dev_free", "bb_dev_download", "bb_dev_upload", "bb_host_malloc", "bb_host_free", "bb_host_malloc_on", "bb_host_free_on",
    "bb_fastq_ingest", "bb_fastq_ingest_dev", "bb_fastq_fetch", "bb_fastq_fetch_lines", "bb_fastq_last_ms",
    "bb_trim_set", "bb_trim_batch", "bb_trim_batch_dev", "bb_trim_plan_dev", "bb_trim_last_ms",
    "bb_format_set_labels", "bb_format_rows_dev",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). barbell_amd has no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64;
    # if ours were loaded first from /opt/rocm, torch would later fail with "No HIP GPUs are
    # available".  Importing torch first makes our DT_NEEDED libamdhip64.so.7 resolve to the copy
    # torch already mapped (same SONAME).  Set BARBELL_AMD_NO_TORCH=1 for a torch-free process.
    if not os.environ.get("BARBELL_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
    L = C.CDLL(SO_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.bb_create.argtypes = [C.POINTER(_abi.GroupDesc), u32, C.POINTER(_abi.Params), C.POINTER(vp)]
    L.bb_create_policy.argtypes = [C.POINTER(_abi.GroupDesc), u32, C.POINTER(_abi.Params), C.POINTER(_abi.Policy), C.POINTER(vp)]
    L.bb_get_policy.argtypes = [vp, C.POINTER(_abi.Policy)]
    L.bb_destroy.argtypes = [vp]
    L.bb_destroy.restype = None
    L.bb_n_groups.argtypes = [vp]
    L.bb_group_get_info.argtypes = [vp, u32, C.POINTER(_abi.GroupInfo)]
    L.bb_group_get_flank.argtypes = [vp, u32, C.c_char_p]
    L.bb_group_get_pattern.argtypes = [vp, u32, u32, i32, C.c_char_p]
    L.bb_annotate_batch.argtypes = [vp, vp, vp, u32, vp, u64, C.POINTER(u64)]
    L.bb_pack_bases.argtypes = [vp, u64, vp]
    L.bb_pack_bases.restype = u64
    L.bb_annotate_batch_packed.argtypes = [vp, vp, vp, vp, u32, vp, u64, C.POINTER(u64)]
    L.bb_annotate_batch_dev.argtypes = [vp, vp, vp, u32, vp, u64, C.POINTER(u64)]
    L.bb_counts_len.argtypes = [vp]
    L.bb_counts_len.restype = u32
    L.bb_counts.argtypes = [vp, vp]
    L.bb_counts_dev.argtypes = [vp]
    L.bb_counts_dev.restype = vp
    L.bb_counts_reset.argtypes = [vp]
    L.bb_last_scan_stats.argtypes = [vp, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    L.bb_filter_twin.argtypes = [vp, u32, C.POINTER(i32), C.POINTER(i32)]
    L.bb_last_length_stats.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.bb_last_host_syncs.argtypes = [vp]
    L.bb_host_phases.argtypes = [vp, i32, vp, vp]
    L.bb_last_barcode_stats.argtypes = [vp, u32, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    L.bb_n_kernels.restype = i32
    L.bb_kernel_name.argtypes = [i32]
    L.bb_kernel_name.restype = C.c_char_p
    L.bb_last_kernel_ms.argtypes = [vp, i32]
    L.bb_last_kernel_ms.restype = C.c_float
    L.bb_set_timing.argtypes = [vp, i32]
    L.bb_set_timing.restype = None
    L.bb_last_dominant_kernel.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_float)]
    L.bb_strerror.argtypes = [i32]
    L.bb_strerror.restype = C.c_char_p
    L.bb_last_error.argtypes = [vp]
    L.bb_last_error.restype = C.c_char_p
    L.bb_build_trace_classes.argtypes = []
    L.bb_build_trace_classes.restype = C.c_uint32
    L.bb_synth_offsets.argtypes = [u64, u32, u32, u64, u32, vp]
    L.bb_synth_reads_host.argtypes = [C.POINTER(_abi.GroupDesc), u32, u64, u32, u32, u64, u32, vp, vp]
    L.bb_synth_reads_dev.argtypes = [vp, u64, u32, u32, u64, u32, vp, vp]
    L.bb_filter_set.argtypes = [vp, vp, u32, vp]
    L.bb_filter_rows.argtypes = [vp, vp, u64, vp]
    L.bb_filter_rows_dev.argtypes = [vp, vp, u64, vp]
    L.bb_inspect_rows.argtypes = [vp, vp, vp, u64, u32, vp]
    L.bb_inspect_rows_dev.argtypes = [vp, vp, vp, u64, u32, vp]
    L.bb_dev_malloc.argtypes = [vp, u64, vp]
    L.bb_dev_free.argtypes = [vp, vp]
    L.bb_dev_free.restype = None
    L.bb_host_malloc.argtypes = [vp, u64, vp]
    L.bb_host_free.argtypes = [vp, vp]
    L.bb_host_free.restype = None
    L.bb_host_malloc_on.argtypes = [i32, u64, vp]
    L.bb_host_free_on.argtypes = [i32, vp]
    L.bb_host_free_on.restype = None
    L.bb_dev_download.argtypes = [vp, vp, vp, u64]
    L.bb_dev_upload.argtypes = [vp, vp, vp, u64]
    L.bb_fastq_ingest.argtypes = [vp, vp, u64, C.c_int, vp, vp]
    L.bb_fastq_ingest_dev.argtypes = [vp, vp, u64, C.c_int, vp, vp]
    L.bb_fastq_fetch.argtypes = [vp] * 8
    L.bb_fastq_last_ms.restype = C.c_float
    L.bb_fastq_last_ms.argtypes = [vp]
    L.bb_trim_set.argtypes = [vp, vp, vp, vp, u32]
    trim_args = [vp, vp, vp, u64, vp, vp, vp, vp, u32, vp, u64, vp, vp, u64, vp, vp, u32, vp, vp]
    L.bb_trim_batch.argtypes = trim_args
    L.bb_trim_batch_dev.argtypes = trim_args
    L.bb_trim_plan_dev.argtypes = [vp, vp, vp, u64, vp, vp, u32, vp, vp, u64, vp, vp, u32, vp, vp]
    L.bb_fastq_fetch_lines.argtypes = [vp, vp]
    L.bb_trim_last_ms.restype = C.c_float
    L.bb_trim_last_ms.argtypes = [vp, C.c_int]
    L.bb_format_set_labels.argtypes = [vp, vp, vp]
    L.bb_format_rows_dev.argtypes = [vp, vp, vp, u64, C.c_int, vp, vp, u64, vp, vp]
    _lib = L
    return L
