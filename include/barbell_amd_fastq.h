/*
 * barbell_amd_fastq.h — C-ABI of FASTQ ingest, the third "next" row of SURVEY.md §8(f): the reference reads
 * its input through paraseq (`open_fastq_collection`, src/io/io.rs:29-33; record loop annotator.rs:245-262,
 * trim.rs:364-384) on the CPU.  Here a block of raw (already decompressed) FASTQ text is parsed on the GPU:
 * newline positions, 4-line records, header split (`split_fastq_header`, io.rs:6-17), and the sequences,
 * qualities and headers are packed into the contiguous batch layout every other entry point of this library
 * takes — the batch is born in HBM and never visits the host as parsed records.
 */
#ifndef BARBELL_AMD_FASTQ_H
#define BARBELL_AMD_FASTQ_H
#include "barbell_amd.h"
#include "barbell_amd_trim.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t n_records;        /* complete 4-line records found                                         */
    uint64_t consumed;         /* bytes of text they span; the caller prepends text[consumed..] to the next block */
    uint64_t n_bases;          /* total sequence bytes (= total quality bytes)                           */
    uint64_t n_hdr;            /* total header bytes (lines without '@')                                 */
    int64_t  bad_record;       /* -1, or the first record that is not FASTQ (see BB_E_FASTQ)             */
} bb_fastq_info;

/* BB_E_FASTQ (barbell_amd.h): a record does not start with '@', its third line not with '+', its sequence
 * and quality lengths differ, or the stream ends inside a record (checked first); bb_fastq_info.bad_record
 * says which.                                                                                          */

/* The parsed batch, in HBM, owned by the context and valid until the next ingest call.  Layout = what
 * bb_annotate_batch_dev (bases, offsets) and bb_trim_batch_dev (quals, headers) take.                   */
typedef struct {
    const uint8_t*  d_bases;
    const uint8_t*  d_quals;
    const uint64_t* d_offsets;     /* n_records + 1 */
    bb_headers      d_headers;     /* four device arrays */
} bb_fastq_batch_dev;

/* `final_block` is a set of flags (0 / 1 as before): */
#define BB_FASTQ_FINAL    1   /* the text is the end of the stream                                                              */
#define BB_FASTQ_TWO_LINE 2   /* compact form: records are their header and sequence lines only ("@id ..\nACGT..\n") — what the
                                 annotate path needs (annotator.rs:125-127 never looks at the quality line), half the bytes to move
                                 over PCIe; the host side drops the '+' and quality lines while it stages the text (host/bb_feed.cpp).
                                 batch->d_quals is NULL for such a block.                                                       */

#define BB_FASTQ_PACKED   4   /* with BB_FASTQ_TWO_LINE: the sequence line holds two bases per byte — what crosses PCIe for annotate is then ~2 KB
                                 per 4-kb read instead of 4.  The kernels only ever look at a read character's IUPAC base set (a 4-bit code:
                                 A=1 C=2 G=4 T=8, their unions, 0 = not an IUPAC letter: matches nothing), so the host keeps exactly that.
                                 Line format: bases 2i, 2i+1 of the line -> byte (code[2i] << 4) | (code[2i+1] ^ 0xA); an odd last base is
                                 paired with code 15; then ONE terminator byte 'E' / 'O' (even / odd number of bases), then '\n'.  No packed
                                 byte is '\n' unless both codes are 0 (two adjacent non-IUPAC characters): the host must not pack such input
                                 (host/bb_annotate.cpp falls back to the plain two-line form).  The ingest unpacks into batch->d_bases as one canonical
                                 character per base set ("-ACMGRSVTWYHKDBN"[code]): rows are those of the original text, the read's own
                                 spelling (case, U for T) is not recoverable — the annotate path never reports it.                          */

/* Parses text[0, text_len).  final_block & BB_FASTQ_FINAL: the text is the end of the stream — a last line without
 * '\n' counts, trailing blank lines are ignored, and a trailing partial record is an error; otherwise
 * the partial tail is left to the caller (info->consumed).  "\r\n" line ends are accepted.
 * The host variant uploads the block first; d_text of the _dev variant must be 16-byte aligned and the allocation
 * must extend at least 15 bytes past text_len (the newline pass reads whole 16-byte pieces; the padding's content
 * does not matter).  Header split: the read id ends at the first `char::is_whitespace` character (ASCII and the
 * UTF-8 encoded Unicode White_Space code points, io.rs:6-17); header bytes are not validated as UTF-8.            */
int bb_fastq_ingest(bb_ctx* ctx, const uint8_t* text, uint64_t text_len, int final_block, bb_fastq_info* info,
                    bb_fastq_batch_dev* batch);
int bb_fastq_ingest_dev(bb_ctx* ctx, const uint8_t* d_text, uint64_t text_len, int final_block, bb_fastq_info* info,
                        bb_fastq_batch_dev* batch);

/* Copies parts of the last ingested batch to the host; any pointer may be NULL (skipped).
 * Sizes: offsets/hdr_offsets n_records+1, id_len/desc_start n_records, hdr n_hdr, bases/quals n_bases. */
int bb_fastq_fetch(bb_ctx* ctx, uint64_t* offsets, uint8_t* hdr, uint64_t* hdr_offsets, uint32_t* id_len, uint32_t* desc_start,
                   uint8_t* bases, uint8_t* quals);

/* Line ends of the last ingested block: line_ends[j] = offset of the '\n' that ends line j of the block's text, j < lines_per_record *
 * n_records (4, or 2 for a BB_FASTQ_TWO_LINE block); record k's header line is text[line_ends[lpr*k-1]+1 .. line_ends[lpr*k]) (from 0
 * for k = 0), a '\r' before the '\n' belongs to the line end.  For a final block whose last line has no '\n' the entry is text_len.
 * With it a caller that still holds the block's text can cut records out of it without parsing it again (bb_trim_plan_dev).        */
int bb_fastq_fetch_lines(bb_ctx* ctx, uint64_t* line_ends);

/* GPU milliseconds of the last ingest (parse + pack, without the upload). */
float bb_fastq_last_ms(bb_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
