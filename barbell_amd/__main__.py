"""`python -m barbell_amd <command>` — the reference's command line (bin/main.rs:17-265) over this package: annotate, filter, trim,
inspect, kit with the reference's flag names and defaults.  The C++ host (`barbell_amd/bin/barbell-amd annotate|kit`) is the fast
front end for FASTQ -> TSV; this one exists so that every documented step of the reference, including the stand-alone ones that start
from an annotation.tsv written earlier, runs on the same files (barbell_amd/steps.py)."""
import argparse
import sys


def _label_space_args(p):
    g = p.add_argument_group("label space (optional: the run's queries; without them the file's own label strings are used)")
    g.add_argument("--kit")
    g.add_argument("--use-extended", action="store_true")
    g.add_argument("-q", "--queries", nargs="+")
    g.add_argument("-b", "--barcode-types", nargs="+", default=["Ftag"])
    p.add_argument("--device", type=int, default=0)


def _barcode_types(names):
    from . import _abi

    out = []
    for s in names:
        if s not in ("Ftag", "Rtag"):
            raise SystemExit(f"Unknown barcode type: {s}, use one of: Ftag, Rtag")   # bin/main.rs:318-322
        out.append(_abi.BB_FTAG if s == "Ftag" else _abi.BB_RTAG)
    return out


def _groups(a, required=False):
    from . import kits

    if getattr(a, "kit", None):
        return kits.groups_from_kit(a.kit, getattr(a, "use_extended", False))
    if getattr(a, "queries", None):
        # the reference's README writes `-q left.fasta,right.fasta -b Ftag,Rtag`, its clap definition takes the values space-separated
        # (bin/main.rs:77-83): both spellings are taken here
        a.queries = [x for v in a.queries for x in v.split(",") if x]
        a.barcode_types = [x for v in a.barcode_types for x in v.split(",") if x]
        types = _barcode_types(a.barcode_types)
        if len(types) != len(a.queries):
            raise SystemExit("--barcode-types must match --queries in number and order")
        return [kits.group_from_fasta(q, t) for q, t in zip(a.queries, types)]
    if required:
        raise SystemExit("--queries is required unless --kit is provided")   # bin/main.rs:311-313
    return None


def parser():
    ap = argparse.ArgumentParser(prog="python -m barbell_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="command", required=True)

    p = sub.add_parser("annotate", help="Annotate FASTQ files with barcode information")
    p.add_argument("-i", "--input", nargs="+", required=True)
    p.add_argument("-t", "--threads", type=int, default=10, help="accepted for compatibility: the records are parsed on the GPU")
    p.add_argument("-o", "--output", default="output.tsv")
    p.add_argument("-q", "--queries", nargs="+")
    p.add_argument("-b", "--barcode-types", nargs="+", default=["Ftag"])
    p.add_argument("--kit")
    p.add_argument("--flank-max-errors", type=int)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--min-score", type=float, default=0.2)
    p.add_argument("--min-score-diff", type=float, default=0.1)
    p.add_argument("--use-extended", action="store_true")
    p.add_argument("--alpha", type=float, default=0.4)
    p.add_argument("--device", type=int, default=0)

    p = sub.add_parser("filter", help="Filter annotation files based on pattern")
    p.add_argument("-i", "--input", required=True)
    p.add_argument("-o", "--output", required=True)
    p.add_argument("-f", "--file", nargs="+", required=True)
    p.add_argument("--dropped")
    p.add_argument("--verbose", action="store_true")
    _label_space_args(p)

    p = sub.add_parser("trim", help="Trim and sort reads based on filtered annotations")
    p.add_argument("-i", "--input", required=True)
    p.add_argument("-r", "--reads", nargs="+", required=True)
    p.add_argument("-o", "--output", required=True)
    p.add_argument("--no-label", action="store_true")
    p.add_argument("--no-orientation", action="store_true")
    p.add_argument("--no-flanks", action="store_true")
    p.add_argument("--sort-labels", action="store_true")
    p.add_argument("--only-side", choices=["left", "right"])
    p.add_argument("--failed-out")
    p.add_argument("--skip-trim", action="store_true")
    p.add_argument("--flip", action="store_true")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--gzip", action="store_true")
    _label_space_args(p)

    p = sub.add_parser("inspect", help="View most common patterns in annotation")
    p.add_argument("-i", "--input", required=True)
    p.add_argument("-n", "--top-n", type=int, default=10)
    p.add_argument("-o", "--read-pattern-out")
    p.add_argument("-s", "--bucket-size", type=int, default=250)
    _label_space_args(p)

    p = sub.add_parser("kit", help="Run a preset")
    p.add_argument("-k", "--kit", required=True)
    p.add_argument("-i", "--input", nargs="+", required=True)
    p.add_argument("-t", "--threads", type=int, default=10, help="accepted for compatibility")
    p.add_argument("-o", "--output", required=True)
    p.add_argument("--maximize", action="store_true")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--min-score", type=float, default=0.2)
    p.add_argument("--min-score-diff", type=float, default=0.1)
    p.add_argument("--flank-max-errors", type=int)
    p.add_argument("--failed-out")
    p.add_argument("--use-extended", action="store_true")
    p.add_argument("--alpha", type=float, default=0.4)
    p.add_argument("--gzip", action="store_true")
    p.add_argument("--device", type=int, default=0)
    return ap


def main(argv=None):
    """0 when the step ran; 1 with 'Error during processing: …' where the reference prints its error and returns (bin/main.rs:300-420)"""
    a = parser().parse_args(argv)
    try:
        return _run(a)
    except (ValueError, OSError, RuntimeError) as e:   # TsvError / pattern errors / BarbellError / unreadable files
        print(f"Error during processing: {e}", file=sys.stderr)
        return 1


def _run(a):
    if a.command == "annotate":
        from . import annotate as A

        if a.kit and a.queries:
            raise SystemExit("--kit conflicts with --queries/--barcode-types")
        print("Starting annotation...")
        total, found = A.annotate(a.input, a.output, _groups(a, required=True), alpha=a.alpha, min_score=a.min_score, min_score_diff=a.min_score_diff,
                                  max_flank_errors=a.flank_max_errors, device=a.device)
        print(f"Annotation complete! {total} reads, {found} with annotations")
        if a.verbose:   # annotator.rs:259-266, :110-112
            import os

            from .steps import write_progress_log

            write_progress_log("annotate", os.path.dirname(a.output), [("Total:", total), ("Kept:", found), ("Dropped:", total - found)])
    elif a.command == "filter":
        from . import steps

        print("Starting filtering...")
        steps.filter_file(a.input, a.output, steps.patterns_from_files(a.file), a.dropped, groups=_groups(a), device=a.device, verbose=a.verbose)
        print("Filtering complete!")
    elif a.command == "trim":
        from . import steps
        from .trim import TrimConfig

        print("Starting trimming...")
        cfg = TrimConfig(not a.no_label, not a.no_orientation, not a.no_flanks, a.sort_labels, a.only_side, a.failed_out, True, a.skip_trim, a.flip,
                         a.verbose, a.gzip)   # bin/main.rs:373-385
        steps.trim_file(a.input, a.reads, a.output, cfg, groups=_groups(a), device=a.device)
        print("Trimming complete!")
    elif a.command == "inspect":
        from . import steps

        print("Inspecting...")
        steps.inspect_file(a.input, a.top_n, a.read_pattern_out, a.bucket_size, groups=_groups(a), device=a.device)
    elif a.command == "kit":
        from .use_kit import demux_using_kit

        demux_using_kit(a.input, a.kit, a.output, maximize=a.maximize, verbose=a.verbose, min_score=a.min_score, min_score_diff=a.min_score_diff,
                        max_flank_errors=a.flank_max_errors, failed_out=a.failed_out, use_extended=a.use_extended, alpha=a.alpha, gzip=a.gzip,
                        device=a.device)
    return 0


if __name__ == "__main__":
    sys.exit(main())
