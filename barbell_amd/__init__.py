"""barbell_amd — MI355X-native annotate hot path of Barbell (HIP kernels behind a C-ABI)."""
from . import _abi, kits  # noqa: F401
from ._abi import ROW_DTYPE  # noqa: F401
from .kits import QueryGroup, group_from_fasta, groups_from_kit  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need the built library
    if name in ("Demuxer", "annotate", "annotate_with_kit", "synth_reads_host", "synth_offsets", "format_rows",
                "TSV_HEADER", "BarbellError"):
        import importlib

        return getattr(importlib.import_module(".annotate", __name__), name)
    raise AttributeError(name)
