"""`barbell kit` — host mirror of src/kits/use_kit.rs:11-109 in one pass over the reads.

The reference runs annotate -> inspect -> filter -> trim as four passes over files (annotation.tsv is
re-read twice, the FASTQ twice).  Here every batch goes to the GPU once; rows, verdicts and reads stay in
HBM for all four steps, and the same files come out: annotation.tsv, pattern_per_read.tsv, filtered.tsv and
one trimmed FASTQ per barcode."""
import os

from . import annotate as A
from . import filter as F
from . import kits
from .inspect_rows import Inspector
from .trim import TrimConfig


def demux_using_kit(fastq_files, kit_name, output_folder, maximize=False, verbose=False, min_score=0.2, min_score_diff=0.1,
                    max_flank_errors=None, failed_out=None, use_extended=False, alpha=0.4, gzip=False, batch_reads=0, device=0,
                    log=print):
    """KitConfig (config.rs:34-48) as keyword arguments.  Returns (total_reads, reads_with_rows, inspector)."""
    os.makedirs(output_folder, exist_ok=True)
    log(f"Kit name: {kit_name}")
    log(f"Kit type: {'Maximize' if maximize else 'Safe'}")
    for tmpl in kits.kit_templates(kit_name):
        log(f"Barcodes: {tmpl['from']} - {tmpl['to']}")
    holder = {}

    def make_inspector(dm):
        holder["i"] = Inspector(dm, os.path.join(output_folder, "pattern_per_read.tsv"), 250)
        return holder["i"]

    stats = {}
    total, found = A.annotate(
        fastq_files, os.path.join(output_folder, "annotation.tsv"), kits.groups_from_kit(kit_name, use_extended), alpha=alpha,
        min_score=min_score, min_score_diff=min_score_diff, max_flank_errors=max_flank_errors, batch_reads=batch_reads, device=device,
        filter_patterns=F.kit_patterns(kit_name, maximize), filtered_file=os.path.join(output_folder, "filtered.tsv"),
        trim_folder=output_folder, trim_config=TrimConfig.for_kit(failed_out, gzip), inspector=make_inspector, stats=stats)
    if verbose:   # use_kit.rs:38,73,97 hands --verbose to its three steps: each leaves its log in the output folder (progress.rs:96-144)
        from .steps import write_progress_log

        write_progress_log("annotate", output_folder, [("Total:", total), ("Kept:", found), ("Dropped:", total - found)])
        write_progress_log("filter", output_folder, [("Total:", stats.get("kept", 0) + stats.get("dropped", 0)), ("Kept:", stats.get("kept", 0)),
                                                     ("Dropped:", stats.get("dropped", 0))])
        write_progress_log("trim", output_folder, [("Total:", total), ("Kept:", stats.get("trimmed", 0)), ("Kept split:", stats.get("split", 0)),
                                                   ("Failed:", stats.get("failed", 0))])
    log("Top 10 most common patterns")
    for line in holder["i"].summary(10):
        log(line)
    return total, found, holder["i"]
