//! src/annotate/gpu.rs — binding of `libbarbell_amd.so` (the MI355X annotate path) for Barbell.
//!
//! New file for the reference tree (rickbeeloo/barbell); add `pub mod gpu;` to `src/annotate/mod.rs`, apply
//! `integration/annotator.patch`, and link with `RUSTFLAGS="-L <dir of libbarbell_amd.so>"` (or a `build.rs` that prints
//! `cargo:rustc-link-search`).  Every item mirrors `include/barbell_amd.h` / `include/barbell_amd_policy.h`; the `#[repr(C)]`
//! layouts are checked field by field against the C headers by `tools/check_rust_layout.py` (offsets from a compiled
//! `offsetof` table, the ctypes mirror and this text must agree).  NOT compiled in the build image of this repository
//! (no Rust toolchain there); it is complete text, not pseudo-code.
//!
//! What it replaces: one `Demuxer` per paraseq worker thread and its per-read `demux` call
//! (`src/annotate/annotator.rs:88-101,123-135`, `src/annotate/searcher.rs:430-490`) by one `GpuDemuxer` per worker and one
//! `annotate_batch` call per paraseq batch.

use crate::annotate::barcodes::{BarcodeGroup, BarcodeType};
use crate::annotate::searcher::BarbellMatch;
use anyhow::{anyhow, Result};
use pa_types::Cost; // as searcher.rs:7 imports it
use sassy::Strand;
use std::ffi::CStr;
use std::os::raw::c_char;

// ---- include/barbell_amd.h ------------------------------------------------------------------------------------------

pub const BB_OK: i32 = 0;
pub const BB_E_CAPACITY: i32 = -7;
pub const BB_FTAG: u8 = 0;
pub const BB_RTAG: u8 = 1;
pub const BB_FFLANK: u8 = 2;
pub const BB_RFLANK: u8 = 3;

#[repr(C)]
pub struct BbGroupDesc {
    pub seqs: *const *const u8,
    pub seq_lens: *const u32,
    pub n_seqs: u32,
    pub type_: u8,
    pub flank_k: i32,
}

#[repr(C)]
pub struct BbParams {
    pub alpha: f32,
    pub min_score: f64,
    pub min_score_diff: f64,
    pub device: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct BbRow {
    pub read_idx: u32,
    pub read_len: u32,
    pub rel_dist_to_end: i32,
    pub read_start_bar: u32,
    pub read_end_bar: u32,
    pub read_start_flank: u32,
    pub read_end_flank: u32,
    pub bar_start: u32,
    pub bar_end: u32,
    pub flank_cost: i16,
    pub barcode_cost: i16,
    pub barcode_idx: i16,
    pub group_idx: u8,
    pub match_type: u8,
    pub strand: u8,
    pub _pad: [u8; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct BbGroupInfo {
    pub flank_len: u32,
    pub prefix_len: u32,
    pub suffix_len: u32,
    pub mask_len: u32,
    pub bar_lo: u32,
    pub bar_hi: u32,
    pub pad_lo: u32,
    pub pad_hi: u32,
    pub pattern_len: u32,
    pub flank_k: i32,
    pub bar_k1: i32,
    pub bar_k2: i32,
    pub perfect_score: f64,
}

// ---- include/barbell_amd_policy.h -----------------------------------------------------------------------------------

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct BbPolicy {
    pub lm_rule: u8,
    pub rc_order: u8,
    pub trace_prio: [u8; 4],
    pub ovh_round: u8,
    pub bar_tie: u8,
    pub lodhi_p: u8,
    pub lodhi_exp: [u8; 4],
    pub rc_path: u8,
    pub _pad: [u8; 2],
    pub lodhi_lambda: f64,
}

impl Default for BbPolicy {
    /// `bb_policy_default`: plateau-right minima, rc matches in scan order, traceback Match > Ins > Sub > Del, floor(alpha * o),
    /// first of equally cheap barcode matches, Lodhi(3, 0.5) with unit decay per alignment column, forward pattern indices on rc paths
    fn default() -> Self {
        BbPolicy {
            lm_rule: 0,
            rc_order: 0,
            trace_prio: [0, 2, 1, 3],
            ovh_round: 0,
            bar_tie: 0,
            lodhi_p: 3,
            lodhi_exp: [1, 1, 1, 1],
            rc_path: 0,
            _pad: [0, 0],
            lodhi_lambda: 0.5,
        }
    }
}

pub enum BbCtx {}

#[link(name = "barbell_amd")]
extern "C" {
    fn bb_create(groups: *const BbGroupDesc, n_groups: u32, params: *const BbParams, out: *mut *mut BbCtx) -> i32;
    fn bb_create_policy(
        groups: *const BbGroupDesc,
        n_groups: u32,
        params: *const BbParams,
        policy: *const BbPolicy,
        out: *mut *mut BbCtx,
    ) -> i32;
    fn bb_destroy(ctx: *mut BbCtx);
    fn bb_group_get_info(ctx: *const BbCtx, group: u32, out: *mut BbGroupInfo) -> i32;
    fn bb_annotate_batch(
        ctx: *mut BbCtx,
        bases: *const u8,
        offsets: *const u64,
        n_reads: u32,
        rows: *mut BbRow,
        rows_cap: u64,
        n_rows: *mut u64,
    ) -> i32;
    fn bb_pack_bases(bases: *const u8, n: u64, out: *mut u8) -> u64;
    fn bb_annotate_batch_packed(
        ctx: *mut BbCtx,
        packed: *const u8,
        packed_offsets: *const u64,
        offsets: *const u64,
        n_reads: u32,
        rows: *mut BbRow,
        rows_cap: u64,
        n_rows: *mut u64,
    ) -> i32;
    fn bb_counts_len(ctx: *const BbCtx) -> u32;
    fn bb_counts(ctx: *mut BbCtx, out: *mut u64) -> i32;
    fn bb_strerror(code: i32) -> *const c_char;
    fn bb_last_error(ctx: *const BbCtx) -> *const c_char;
}

fn c_text(p: *const c_char) -> String {
    if p.is_null() {
        return String::new();
    }
    // the library returns NUL-terminated text that lives as long as the context / the thread
    unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
}

fn error_text(code: i32, ctx: *const BbCtx) -> anyhow::Error {
    let what = c_text(unsafe { bb_strerror(code) });
    let detail = c_text(unsafe { bb_last_error(ctx) });
    if detail.is_empty() {
        anyhow!("barbell_amd: {what} ({code})")
    } else {
        anyhow!("barbell_amd: {what} ({code}): {detail}")
    }
}

/// The query sequences `BarcodeGroup::new` was given (`barcodes.rs:105-197`), rebuilt from what the group keeps: the library does its
/// own query preparation from the original `<prefix><barcode><suffix>` strings.  `Barcode::seq` is `seq[pad_start..min(pad_end, len)]`,
/// so the barcode proper sits at `prefix_len - pad_start` and is `bar_region.1 - bar_region.0 + 1` long.
pub fn original_queries(group: &BarcodeGroup) -> Vec<Vec<u8>> {
    let prefix_len = group.flank_prefix.len();
    let core_at = prefix_len - group.pad_region.0;
    let mask_len = group.bar_region.1 - group.bar_region.0 + 1;
    group
        .barcodes
        .iter()
        .map(|b| {
            let mut q = Vec::with_capacity(prefix_len + mask_len + group.flank_suffix.len());
            q.extend_from_slice(&group.flank_prefix);
            q.extend_from_slice(&b.seq[core_at..core_at + mask_len]);
            q.extend_from_slice(&group.flank_suffix);
            q
        })
        .collect()
}

fn match_type_code(t: &BarcodeType) -> Result<u8> {
    match t {
        BarcodeType::Ftag => Ok(BB_FTAG),
        BarcodeType::Rtag => Ok(BB_RTAG),
        other => Err(anyhow!("a query group must be Ftag or Rtag, not {}", other.as_str())),
    }
}

fn match_type_of(code: u8) -> Result<BarcodeType> {
    match code {
        BB_FTAG => Ok(BarcodeType::Ftag),
        BB_RTAG => Ok(BarcodeType::Rtag),
        BB_FFLANK => Ok(BarcodeType::Fflank),
        BB_RFLANK => Ok(BarcodeType::Rflank),
        other => Err(anyhow!("barbell_amd: row with match_type {other}")),
    }
}

/// Reads per GPU call.  paraseq hands a worker ~1 k records at a time (`process_parallel(.., None)`, `annotator.rs:278-280`); a call into the
/// library costs ~0.4 ms of its thread whatever it carries, so `DemuxProcessor` collects batches until it holds this many reads (32 MB of
/// bases, 16 MB packed, per worker).  Measured on one MI355X with the reference's ten worker threads (`bench.py` -> `boundary_step`):
/// calls of 1 024 reads 7.7 M reads/s, of 8 192 reads 12.7 M (PCIe-bound at 4 KB per read), of 8 192 packed reads 15-21 M.
pub const GPU_BATCH_READS: usize = 8192;

/// Appends `seq` two bases per byte (`bb_pack_bases`: 4-bit IUPAC base sets; AVX2 / AVX-512 inside the library) — the form
/// `GpuDemuxer::annotate_batch_packed` takes.  Every read begins at a byte of its own.
pub fn pack_bases_into(seq: &[u8], out: &mut Vec<u8>) {
    let at = out.len();
    let n = (seq.len() + 1) / 2;
    out.resize(at + n, 0);
    let wrote = unsafe { bb_pack_bases(seq.as_ptr(), seq.len() as u64, out.as_mut_ptr().add(at)) };
    debug_assert_eq!(wrote as usize, n);
}

/// One annotate context on one GPU: the stand-in for `Demuxer` (`searcher.rs:202-226`).  Not `Sync`, like `Demuxer` (`&mut self`);
/// one per paraseq worker thread, `device = thread_id % n_gpus`.
pub struct GpuDemuxer {
    ctx: *mut BbCtx,
    rows: Vec<BbRow>,
}

// the context owns its HIP stream and buffers; it may move to the worker thread that uses it
unsafe impl Send for GpuDemuxer {}

impl GpuDemuxer {
    /// `Demuxer::new(alpha, verbose, min_score, min_score_diff)` + `add_query_group` for every group.  Reference panics
    /// (`barcodes.rs:113-147`) come back as errors.  `policy = None`: `$BARBELL_AMD_POLICY` or the library's default.
    pub fn new(
        groups: &[BarcodeGroup],
        alpha: f32,
        min_score: f64,
        min_score_diff: f64,
        device: i32,
        policy: Option<&BbPolicy>,
    ) -> Result<Self> {
        let queries: Vec<Vec<Vec<u8>>> = groups.iter().map(original_queries).collect();
        let ptrs: Vec<Vec<*const u8>> =
            queries.iter().map(|qs| qs.iter().map(|q| q.as_ptr()).collect()).collect();
        let lens: Vec<Vec<u32>> = queries.iter().map(|qs| qs.iter().map(|q| q.len() as u32).collect()).collect();
        let mut descs = Vec::with_capacity(groups.len());
        for (i, g) in groups.iter().enumerate() {
            descs.push(BbGroupDesc {
                seqs: ptrs[i].as_ptr(),
                seq_lens: lens[i].as_ptr(),
                n_seqs: queries[i].len() as u32,
                type_: match_type_code(&g.barcode_type)?,
                flank_k: g.k_cutoff.map(|k| k as i32).unwrap_or(-1), // -1: the automatic cutoff (edit_model.rs:2-11)
            });
        }
        let params = BbParams { alpha, min_score, min_score_diff, device };
        let mut ctx: *mut BbCtx = std::ptr::null_mut();
        let rc = unsafe {
            match policy {
                Some(p) => bb_create_policy(descs.as_ptr(), descs.len() as u32, &params, p, &mut ctx),
                None => bb_create(descs.as_ptr(), descs.len() as u32, &params, &mut ctx),
            }
        };
        if rc != BB_OK {
            return Err(error_text(rc, std::ptr::null()));
        }
        Ok(GpuDemuxer { ctx, rows: Vec::new() })
    }

    /// Geometry the library derived for group `g` (the public fields of `BarcodeGroup`, `barcodes.rs:57-72`).
    pub fn group_info(&self, g: usize) -> Result<BbGroupInfo> {
        let mut info = BbGroupInfo::default();
        let rc = unsafe { bb_group_get_info(self.ctx, g as u32, &mut info) };
        if rc != BB_OK {
            return Err(error_text(rc, self.ctx));
        }
        Ok(info)
    }

    /// `demux` for every read of a batch (`searcher.rs:430-490` incl. `collapse_overlapping_matches`): `bases` are the reads' bytes
    /// back to back exactly as in the FASTQ, `offsets[i]..offsets[i + 1]` the bytes of read i (`offsets.len()` = reads + 1).
    /// Rows come back in read order, rows of one read in `read_start_flank` order.
    pub fn annotate_batch(&mut self, bases: &[u8], offsets: &[u64]) -> Result<&[BbRow]> {
        let n_reads = (offsets.len().max(1) - 1) as u32;
        if self.rows.len() < 4 * n_reads as usize + 64 {
            self.rows.resize(4 * n_reads as usize + 64, BbRow::default());
        }
        let mut n_rows = 0u64;
        let mut rc = unsafe {
            bb_annotate_batch(
                self.ctx,
                bases.as_ptr(),
                offsets.as_ptr(),
                n_reads,
                self.rows.as_mut_ptr(),
                self.rows.len() as u64,
                &mut n_rows,
            )
        };
        if rc == BB_E_CAPACITY {
            // n_rows holds the number of rows the batch has: grow once and run it again
            self.rows.resize(n_rows as usize, BbRow::default());
            rc = unsafe {
                bb_annotate_batch(
                    self.ctx,
                    bases.as_ptr(),
                    offsets.as_ptr(),
                    n_reads,
                    self.rows.as_mut_ptr(),
                    self.rows.len() as u64,
                    &mut n_rows,
                )
            };
        }
        if rc != BB_OK {
            return Err(error_text(rc, self.ctx));
        }
        Ok(&self.rows[..n_rows as usize])
    }

    /// The same for reads packed with `pack_bases_into`: read i is `packed[packed_offsets[i]..packed_offsets[i + 1]]`, its bases
    /// `offsets[i]..offsets[i + 1]` (all three vectors as `DemuxProcessor` grows them record by record).  Rows are those of
    /// `annotate_batch` on the original bytes; half the bytes cross PCIe.
    pub fn annotate_batch_packed(&mut self, packed: &[u8], packed_offsets: &[u64], offsets: &[u64]) -> Result<&[BbRow]> {
        let n_reads = (offsets.len().max(1) - 1) as u32;
        if packed_offsets.len() != offsets.len() {
            return Err(anyhow!("barbell_amd: {} packed offsets for {} offsets", packed_offsets.len(), offsets.len()));
        }
        if self.rows.len() < 4 * n_reads as usize + 64 {
            self.rows.resize(4 * n_reads as usize + 64, BbRow::default());
        }
        let mut n_rows = 0u64;
        let mut rc = BB_E_CAPACITY;
        for _attempt in 0..2 {
            rc = unsafe {
                bb_annotate_batch_packed(
                    self.ctx,
                    packed.as_ptr(),
                    packed_offsets.as_ptr(),
                    offsets.as_ptr(),
                    n_reads,
                    self.rows.as_mut_ptr(),
                    self.rows.len() as u64,
                    &mut n_rows,
                )
            };
            if rc != BB_E_CAPACITY {
                break;
            }
            // n_rows holds the number of rows the batch has: grow once and run it again
            self.rows.resize(n_rows as usize, BbRow::default());
        }
        if rc != BB_OK {
            return Err(error_text(rc, self.ctx));
        }
        Ok(&self.rows[..n_rows as usize])
    }

    /// Rows per (group, barcode | flank-only) since the context was made: per group its barcodes in order, then one flank-only slot.
    pub fn counts(&mut self) -> Result<Vec<u64>> {
        let mut out = vec![0u64; unsafe { bb_counts_len(self.ctx) } as usize];
        let rc = unsafe { bb_counts(self.ctx, out.as_mut_ptr()) };
        if rc != BB_OK {
            return Err(error_text(rc, self.ctx));
        }
        Ok(out)
    }
}

impl Drop for GpuDemuxer {
    fn drop(&mut self) {
        unsafe { bb_destroy(self.ctx) }
    }
}

/// `bb_row` -> `BarbellMatch` (`searcher.rs:31-64`): the strings the row leaves out come from the caller's read ids and groups.
pub fn to_barbell_match(row: &BbRow, read_ids: &[String], groups: &[BarcodeGroup]) -> Result<BarbellMatch> {
    let group = groups
        .get(row.group_idx as usize)
        .ok_or_else(|| anyhow!("barbell_amd: row of group {} of {}", row.group_idx, groups.len()))?;
    let label = if row.barcode_idx < 0 {
        "flank".to_string() // searcher.rs:258
    } else {
        group.barcodes[row.barcode_idx as usize].label.clone()
    };
    Ok(BarbellMatch::new(
        row.read_start_bar as usize,
        row.read_end_bar as usize,
        row.read_start_flank as usize,
        row.read_end_flank as usize,
        row.bar_start as usize,
        row.bar_end as usize,
        match_type_of(row.match_type)?,
        row.flank_cost as Cost,
        row.barcode_cost as Cost,
        label,
        if row.strand == 0 { Strand::Fwd } else { Strand::Rc },
        row.read_len as usize,
        read_ids[row.read_idx as usize].clone(),
        row.rel_dist_to_end as isize,
        None,
    ))
}

/// Number of different reads among rows that are sorted by read (what `found_count` counts, `annotator.rs:129-132`).
pub fn reads_with_rows(rows: &[BbRow]) -> usize {
    let mut n = 0usize;
    let mut last = u32::MAX;
    for r in rows {
        if r.read_idx != last {
            n += 1;
            last = r.read_idx;
        }
    }
    n
}

#[cfg(test)]
mod test {
    use super::*;

    // the layouts the C side was compiled with (include/barbell_amd.h: "bb_row is 48 bytes", bb_policy "24 bytes")
    #[test]
    fn layouts() {
        assert_eq!(std::mem::size_of::<BbRow>(), 48);
        assert_eq!(std::mem::size_of::<BbPolicy>(), 24);
        assert_eq!(std::mem::size_of::<BbParams>(), 32);
        assert_eq!(std::mem::size_of::<BbGroupDesc>(), 32);
        assert_eq!(std::mem::size_of::<BbGroupInfo>(), 56);
    }

    #[test]
    fn read_counting() {
        let mut a = BbRow::default();
        let mut b = BbRow::default();
        a.read_idx = 3;
        b.read_idx = 5;
        assert_eq!(reads_with_rows(&[a, a, b]), 2);
        assert_eq!(reads_with_rows(&[]), 0);
    }
}
