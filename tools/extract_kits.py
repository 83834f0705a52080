#!/usr/bin/env python3
"""Extracts the kit DATA tables (flank constants, templates, barcode sequences, kit-name map) of
the reference (src/kits/kits.rs:9-47,252-464,635-708,819-1103) into barbell_amd/data/kits.json.
Only data is extracted — no reference code.  Run in the authoring container:
    python tools/extract_kits.py /root/reference
"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = open(os.path.join(ref, "src/kits/kits.rs")).read()

consts = dict(re.findall(r'const\s+([A-Z0-9_]+):\s*&str\s*=\s*"([A-Za-z]*)";', src))

arrays = {}
for name, body in re.findall(r'const\s+([A-Z_]+_SEQS):\s*\[&str;\s*\d+\]\s*=\s*\[(.*?)\];', src, re.S):
    arrays[name] = re.findall(r'"([ACGT]+)"', body)

rbk_special = {int(n): s for n, s in re.findall(r'(\d+)\s*=>\s*Some\("([ACGT]+)"\)', src)}

templates = {}
for name, body in re.findall(r'static\s+(TEMPLATES_[A-Z0-9_]+):\s*&\[TemplateSpec\]\s*=\s*&\[(.*?)\];', src, re.S):
    body = "\n".join(l for l in body.splitlines() if not l.strip().startswith("//"))
    specs = []
    for spec in re.findall(r'TemplateSpec\s*\{(.*?template_type:[^\n]*)\n', body, re.S):
        parts = re.search(r'parts:\s*&\[(.*?)\]', spec, re.S).group(1)
        parts = [p.strip().strip('"') for p in parts.split(",") if p.strip()]
        parts = [consts.get(p, p) for p in parts]
        lr = re.search(r'LabelRange::(new|new_12a)\("(\w+)",\s*"(\w+)"\)', spec)
        specs.append({
            "parts": parts,
            "from": lr.group(2), "to": lr.group(3), "use_12a": lr.group(1) == "new_12a",
            "side": re.search(r'TemplateBarcodeType::(\w+)', spec).group(1),
            "type": re.search(r'TemplateType::(\w+)', spec).group(1),
        })
    templates[name] = specs

kitconst = {}
for name, body in re.findall(r'const\s+(KIT_[A-Z0-9_]+):\s*KitConfig\s*=\s*KitConfig::new\((.*?)\);', src, re.S):
    kitconst[name] = re.search(r'(TEMPLATES_[A-Z0-9_]+)', body).group(1)

kits = {}
fn = re.search(r'pub fn get_kit_info.*?\n\}', src, re.S).group(0)
kit_of_const = {}
for kit, const in re.findall(r'"([A-Z0-9\-]+)"\s*=>\s*(KIT_[A-Z0-9_]+)', fn):
    kits[kit] = kitconst[const]
    kit_of_const[kit] = const

# filter pattern sets (kits.rs:175-236) and which set each kit uses (KitConfig::new(.., safe_patterns, maximize_patterns, ..))
pattern_sets = {}
for name, body in re.findall(r'static\s+([A-Z_]+_PATTERNS_[A-Z]+):\s*LazyLock<Vec<Pattern>>\s*=\s*LazyLock::new\(\|\|\s*\{\s*vec!\[(.*?)\n\s*\]\s*\}\);', src, re.S):
    pattern_sets[name] = re.findall(r'pattern_from_str!\(\s*"(.*?)"\s*\)', body, re.S)
fns = dict(re.findall(r"fn\s+(\w+)\(\)\s*->\s*&'static \[Pattern\]\s*\{\s*&(\w+)\s*\}", src))
kit_patterns = {}
for name, body in re.findall(r'const\s+(KIT_[A-Z0-9_]+):\s*KitConfig\s*=\s*KitConfig::new\((.*?)\);', src, re.S):
    parts = [p.strip() for p in body.split(',') if p.strip()]
    kit_patterns[name] = {"safe": fns[parts[2]], "maximize": fns[parts[3]]}
kit_filter = {kit: kit_patterns[const] for kit, const in kit_of_const.items()}

out = {
    "_source": "rickbeeloo/barbell v0.3.3 src/kits/kits.rs (data tables only), extracted by tools/extract_kits.py",
    "seqs": {"BC": arrays["BC_SEQS"], "NB": arrays["NB_SEQS"], "AB": arrays["AB_SEQS"], "BP": arrays["BP_SEQS"]},
    "BC12A": consts["BC12A_SEQ"],
    "RBK_special": rbk_special,
    "templates": templates,
    "kits": kits,
    "pattern_sets": pattern_sets,
    "kit_filter": kit_filter,
}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "barbell_amd", "data", "kits.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, len(kits), "kits", {k: len(v) for k, v in out["seqs"].items()})
