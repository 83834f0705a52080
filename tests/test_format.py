"""TSV lines rendered on the GPU (include/barbell_amd_format.h) against the host-side statement of the same format
(barbell_amd.annotate.format_rows = the csv-crate serialisation of BarbellMatch, searcher.rs:31-142): byte-exact for
annotation.tsv, filtered.tsv and the dropped file, including read ids the csv writer has to quote."""
import numpy as np
import pytest

from tests.common import config_groups

pytestmark = pytest.mark.gpu


def fastq_text(ids, bases, offsets):
    out = []
    for i, rid in enumerate(ids):
        s = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
        out.append(b"@" + rid + b" ch=7 note\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("cfg", ["nbd96", "dual", "rbk24"])
def test_rendered_lines_equal_host_format(cfg):
    from barbell_amd import annotate as A
    from barbell_amd import fastq as Q
    from barbell_amd import filter as F
    from barbell_amd.format import FMT_DROPPED, FMT_KEPT, RowFormatter

    groups = config_groups(cfg)
    n = 700
    bases, offsets = A.synth_reads_host(groups, 77, 300, 2500, 0, n)
    ids = [b"read_%d" % i for i in range(n)]
    ids[3] = b'q"uo"ted'          # the csv writer quotes this one and doubles its quotes
    ids[5] = b'"'
    ids[9] = b"x" * 70
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    info, batch = Q.ingest(dm, fastq_text(ids, bases, offsets), True)
    assert int(info.n_records) == n
    rows = dm.demux_ingested(batch, n)
    assert len(rows) > n // 2
    d_rows = dm.buf("rows").ptr
    if cfg == "dual":  # labels come from FASTA headers: one the csv writer has to quote (delimiter, quote), like the ids above
        tags = rows[(rows["group_idx"] == 0) & (rows["barcode_idx"] >= 0)]["barcode_idx"]
        groups[0].labels[int(np.bincount(tags).argmax())] = 'bar"code\t7'
    fmt = RowFormatter(dm, groups)
    sids = [i.decode() for i in ids]
    text, nl = fmt.render(d_rows, len(rows), batch)
    want = "".join(l + "\n" for l in A.format_rows(rows, sids, groups)).encode()
    assert nl == len(rows) and text == want
    assert cfg != "dual" or b'\t"bar""code\t7"\t' in text
    pats = F.kit_patterns("SQK-NBD114-96", True) if cfg == "nbd96" else [F.pattern_from_str("Ftag[fw, *, @left(0..250), >>]"),
                                                                           F.pattern_from_str("Ftag[fw, *, @left(0..250), >>]__Rtag[<<, fw, *, @right(0..250)]")]
    flt = F.Filter(dm, pats)
    v = flt.verdicts_ingested(d_rows, len(rows))
    d_v = dm.buf("verdicts").ptr
    keep = v["pass"] == 1
    assert keep.any() and (~keep).any()
    for mode, sel in ((FMT_KEPT, keep), (FMT_DROPPED, ~keep)):
        text, nl = fmt.render(d_rows, len(rows), batch, mode, d_v)
        want = "".join(l + "\n" for l in A.format_rows(rows[sel], sids, groups, v[sel])).encode()
        assert nl == int(sel.sum()) and text == want
    assert b"After(" in fmt.render(d_rows, len(rows), batch, FMT_KEPT, d_v)[0]
    # empty batch / capacity protocol
    assert fmt.render(d_rows, 0, batch) == (b"", 0)
