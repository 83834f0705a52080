// kat.rs — prints known-answer vectors of the three third-party primitives that decide Barbell's barcode call,
// one JSON object per line, in the format tests/test_ref_golden.py ingests (tests/golden/ref_kat.jsonl):
//   {"kind":"lodhi","ops":"===X=I=D","score":1.25,"bits":"0x3ff4000000000000"}
//   {"kind":"search","searcher":"rc"|"rc_overhang","alpha":0.4,"pattern":"..","text":"..","k":5,"matches":[M..]}
//   {"kind":"search_set","patterns":["..",..],"text":"..","k":16,"matches":[M..]}       (M carries pattern_idx)
//   M = {"text_start","text_end","pattern_start","pattern_end","cost","strand":"Fwd"|"Rc","pattern_idx","ops","path":[[i,j],..]}
// Call sites restated from Barbell v0.3.3: searcher.rs:209-211 (Lodhi::new(3, 0.5), new_rc_with_overhang, new_rc),
// :282-288 (search_encoded_patterns), :367 (lodhi.compute), :438 (overhang search); barcodes.rs:89-90 (encode_patterns).
// UNCOMPILED in this repo's image; field/method names follow the reference's own uses of these crates.
use cigar_lodhi_rs::*;
use pa_types::*;
use sassy::profiles::Iupac;
use sassy::{Match, Searcher, Strand};

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
fn lcg(state: &mut u64) -> u64 { *state = state.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407); *state >> 33 }
fn rand_seq(state: &mut u64, n: usize) -> Vec<u8> { (0..n).map(|_| b"ACGT"[(lcg(state) & 3) as usize]).collect() }
fn mutate(state: &mut u64, s: &[u8], n_edits: usize) -> Vec<u8> {
    let mut v = s.to_vec();
    for _ in 0..n_edits {
        if v.is_empty() { break; }
        let p = (lcg(state) as usize) % v.len();
        match lcg(state) % 3 { 0 => v[p] = b"ACGT"[(lcg(state) & 3) as usize], 1 => { v.insert(p, b"ACGT"[(lcg(state) & 3) as usize]); } _ => { v.remove(p); } }
    }
    v
}

fn main() {
    let mut st: u64 = 0xBA7BE11;
    // ---- H8: Lodhi::new(3, 0.5).compute on ~24 CIGARs: all-match lengths (the normaliser of searcher.rs:229-239),
    // single edits of each kind at different positions, runs, and random op strings
    let mut lodhi = Lodhi::new(3, 0.5);
    let mut cigars: Vec<String> = vec!["=".repeat(3), "=".repeat(4), "=".repeat(10), "=".repeat(41), "=".repeat(42), "=".repeat(44),
        "==X==".into(), "==I==".into(), "==D==".into(), "=X=X=X=".into(), "====DDDD====".into(), "====IIII====".into(),
        "X===".into(), "===X".into(), "D=====".into(), "=====D".into(), "=".repeat(20) + "X" + &"=".repeat(21), "=".repeat(10) + "ID" + &"=".repeat(30)];
    for _ in 0..8 { let n = 30 + (lcg(&mut st) % 20) as usize; cigars.push((0..n).map(|_| ['=', '=', '=', 'X', 'I', 'D'][(lcg(&mut st) % 6) as usize]).collect()); }
    for ops in &cigars {
        let v = lodhi.compute(&cigar_of(ops));
        println!("{{\"kind\":\"lodhi\",\"ops\":\"{}\",\"score\":{:e},\"bits\":\"{:#018x}\"}}", ops, v, v.to_bits());
    }
    // ---- H1-H6: Searcher::search, with and without overhang, on the NBD114-96 masked flank and a 90-nt RBK-like flank
    let flanks: [&[u8]; 2] = [b"ATTGCTAAGGTTAANNNNNNNNNNNNNNNNNNNNNNNNCAGCACCT",
                              b"GCTTGGGTGTTTAACCNNNNNNNNNNNNNNNNNNNNNNNNGTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"];
    for (fi, flank) in flanks.iter().enumerate() {
        for case in 0..12 {
            let k = if fi == 0 { [0usize, 3, 5][case % 3] } else { [5usize, 20][case % 2] };
            let solid: Vec<u8> = flank.iter().map(|&c| if c == b'N' { b"ACGT"[(lcg(&mut st) & 3) as usize] } else { c }).collect();
            let inst = mutate(&mut st, &solid, case % 5);
            let lead = (lcg(&mut st) % 40) as usize;
            let mut text = rand_seq(&mut st, lead);
            match case % 4 {
                0 => { text.extend_from_slice(&inst); text.extend(rand_seq(&mut st, 120)); }                      // near the 5' end
                1 => { text = inst[inst.len() / 3..].to_vec(); text.extend(rand_seq(&mut st, 150)); }             // truncated: left overhang
                2 => { text.extend(rand_seq(&mut st, 100)); text.extend_from_slice(&inst[..inst.len() * 2 / 3]); } // right overhang
                _ => { text.extend_from_slice(&inst); text.extend(rand_seq(&mut st, 60));                         // + an rc copy downstream
                       let rc: Vec<u8> = inst.iter().rev().map(|&c| match c { b'A' => b'T', b'C' => b'G', b'G' => b'C', _ => b'A' }).collect(); text.extend(rc); text.extend(rand_seq(&mut st, 30)); }
            }
            for (name, alpha) in [("rc", -1.0f32), ("rc_overhang", 0.4f32), ("rc_overhang", 1.0f32)] {
                let mut s = if alpha < 0.0 { Searcher::<Iupac>::new_rc() } else { Searcher::<Iupac>::new_rc_with_overhang(alpha) };
                let ms = s.search(*flank, &text, k);
                let js: Vec<String> = ms.iter().map(json_match).collect();
                println!("{{\"kind\":\"search\",\"searcher\":\"{}\",\"alpha\":{},\"pattern\":\"{}\",\"text\":\"{}\",\"k\":{},\"matches\":[{}]}}",
                    name, alpha, String::from_utf8_lossy(flank), String::from_utf8_lossy(&text), k, js.join(","));
            }
        }
    }
    // ---- H7: search_encoded_patterns on sets of equally long padded barcodes against ~45-nt windows, k = 0.4 m and k = m
    for case in 0..10 {
        let m = [42usize, 44, 41][case % 3];
        let left = rand_seq(&mut st, 10); let right = rand_seq(&mut st, m - 34);
        let pats: Vec<Vec<u8>> = (0..12).map(|_| { let mut p = left.clone(); p.extend(rand_seq(&mut st, 24)); p.extend_from_slice(&right); p }).collect();
        let pick = (lcg(&mut st) as usize) % pats.len();
        let mut win = rand_seq(&mut st, (lcg(&mut st) % 4) as usize);
        win.extend(mutate(&mut st, &pats[pick], case % 6));
        win.extend(rand_seq(&mut st, (lcg(&mut st) % 4) as usize));
        let enc = Searcher::<Iupac>::new_fwd().encode_patterns(&pats);
        for k in [(m as f32 * 0.4) as usize, m] {
            let mut s = Searcher::<Iupac>::new_rc();
            let js: Vec<String> = s.search_encoded_patterns(&enc, &win, k).iter().map(json_match).collect();
            let ps: Vec<String> = pats.iter().map(|p| format!("\"{}\"", String::from_utf8_lossy(p))).collect();
            println!("{{\"kind\":\"search_set\",\"patterns\":[{}],\"text\":\"{}\",\"k\":{},\"matches\":[{}]}}", ps.join(","), String::from_utf8_lossy(&win), k, js.join(","));
        }
    }
}
