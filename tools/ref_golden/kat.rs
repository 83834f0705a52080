// kat.rs — feeds tools/ref_golden/kat_inputs.tsv to the REAL crates Barbell v0.3.3 pins (sassy 0.2.1, cigar-lodhi-rs 0.1.0,
// pa-types 1.2.0) and prints their answers, one JSON object per line, in the format tests/test_ref_golden.py and
// tools/ref_fit.py ingest (tests/golden/ref_kat.jsonl):
//   {"kind":"lodhi","ops":"===X=I=D","score":1.25,"bits":"0x3ff4000000000000"}
//   {"kind":"search","searcher":"rc"|"rc_overhang","alpha":0.4,"pattern":"..","text":"..","k":5,"matches":[M..]}
//   {"kind":"search_set","patterns":["..",..],"text":"..","k":16,"matches":[M..]}       (M carries pattern_idx)
//   M = {"text_start","text_end","pattern_start","pattern_end","cost","strand":"Fwd"|"Rc","pattern_idx","ops","path":[[i,j],..]}
// The inputs (written by gen_kat_inputs.py) hold, for every switchable assumption of include/barbell_amd_policy.h, cases
// on which its alternatives answer differently, so `tools/ref_fit.py kat ref_kat.jsonl` can name the policy the crates follow.
// Call sites restated from Barbell v0.3.3: searcher.rs:209-211 (Lodhi::new(3, 0.5), new_rc_with_overhang, new_rc),
// :282-288 (search_encoded_patterns), :367 (lodhi.compute), :438 (overhang search); barcodes.rs:89-90 (encode_patterns).
// UNCOMPILED in this repo's image (no Rust toolchain); field/method names follow the reference's own uses of these crates.
//   cd tools/ref_golden && cargo run --release > ../../tests/golden/ref_kat.jsonl
use cigar_lodhi_rs::*;
use pa_types::*;
use sassy::profiles::Iupac;
use sassy::{Match, Searcher, Strand};
use std::io::BufRead;

fn ops_of(c: &Cigar) -> String {
    let mut s = String::new();
    for el in &c.ops {
        let ch = match el.op { CigarOp::Match => '=', CigarOp::Sub => 'X', CigarOp::Ins => 'I', CigarOp::Del => 'D' };
        for _ in 0..el.cnt { s.push(ch); }
    }
    s
}
fn cigar_of(ops: &str) -> Cigar {
    let mut v: Vec<CigarElem> = Vec::new();
    for ch in ops.chars() {
        let op = match ch { '=' => CigarOp::Match, 'X' => CigarOp::Sub, 'I' => CigarOp::Ins, _ => CigarOp::Del };
        match v.last_mut() { Some(l) if l.op == op => l.cnt += 1, _ => v.push(CigarElem { op, cnt: 1 }) }
    }
    Cigar { ops: v }
}
fn json_match(m: &Match) -> String {
    let path: Vec<String> = m.to_path().iter().map(|Pos(i, j)| format!("[{},{}]", i, j)).collect();
    format!("{{\"text_start\":{},\"text_end\":{},\"pattern_start\":{},\"pattern_end\":{},\"cost\":{},\"strand\":\"{}\",\"pattern_idx\":{},\"ops\":\"{}\",\"path\":[{}]}}",
        m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost,
        match m.strand { Strand::Fwd => "Fwd", Strand::Rc => "Rc" }, m.pattern_idx, ops_of(&m.cigar), path.join(","))
}

fn main() {
    let path = std::env::args().nth(1).unwrap_or_else(|| "kat_inputs.tsv".to_string());
    let file = std::fs::File::open(&path).expect("kat_inputs.tsv (written by gen_kat_inputs.py)");
    let mut lodhi = Lodhi::new(3, 0.5);                                               // searcher.rs:209
    for line in std::io::BufReader::new(file).lines() {
        let line = line.unwrap();
        if line.is_empty() || line.starts_with('#') { continue; }
        let f: Vec<&str> = line.split('\t').collect();
        match f[0] {
            "lodhi" => {                                                              // searcher.rs:367
                let v = lodhi.compute(&cigar_of(f[1]));
                println!("{{\"kind\":\"lodhi\",\"ops\":\"{}\",\"score\":{:e},\"bits\":\"{:#018x}\"}}", f[1], v, v.to_bits());
            }
            "search" => {                                                             // searcher.rs:210-211, 438
                let alpha: f32 = f[1].parse().unwrap();
                let k: usize = f[2].parse().unwrap();
                let (pattern, text) = (f[3].as_bytes(), f[4].as_bytes());
                let mut s = if alpha < 0.0 { Searcher::<Iupac>::new_rc() } else { Searcher::<Iupac>::new_rc_with_overhang(alpha) };
                let js: Vec<String> = s.search(pattern, &text, k).iter().map(json_match).collect();
                println!("{{\"kind\":\"search\",\"searcher\":\"{}\",\"alpha\":{},\"pattern\":\"{}\",\"text\":\"{}\",\"k\":{},\"matches\":[{}]}}",
                    if alpha < 0.0 { "rc" } else { "rc_overhang" }, alpha, f[3], f[4], k, js.join(","));
            }
            _ => {                                                                    // barcodes.rs:89-90, searcher.rs:282-288
                let k: usize = f[1].parse().unwrap();
                let text = f[2].as_bytes();
                let pats: Vec<Vec<u8>> = f[3].split(',').map(|p| p.as_bytes().to_vec()).collect();
                let enc = Searcher::<Iupac>::new_fwd().encode_patterns(&pats);
                let mut s = Searcher::<Iupac>::new_rc();
                let js: Vec<String> = s.search_encoded_patterns(&enc, &text, k).iter().map(json_match).collect();
                let ps: Vec<String> = f[3].split(',').map(|p| format!("\"{}\"", p)).collect();
                println!("{{\"kind\":\"search_set\",\"patterns\":[{}],\"text\":\"{}\",\"k\":{},\"matches\":[{}]}}", ps.join(","), f[2], k, js.join(","));
            }
        }
    }
}
