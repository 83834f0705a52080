#!/usr/bin/env python3
"""check_rust_layout.py — integration/gpu.rs against include/*.h.

The Rust binding cannot be compiled in this image (no toolchain), so its `#[repr(C)]` structs and `extern "C"` block are checked as TEXT:
  * every struct: field names in order, every field's offset and size and the struct's size under the repr(C) rules, against
    `offsetof` / `sizeof` printed by a C program compiled here from the real headers (gcc), and against the ctypes mirror
    (barbell_amd/_abi.py) the tests drive the library through;
  * every `fn bb_*` of the extern block: declared in include/barbell_amd.h with the same number of parameters, same return kind;
  * constants (BB_E_CAPACITY, BB_FTAG ...) equal the headers' values;
  * no pseudo-code markers left in integration/ ("/* retry */", "/* distinct", "todo!", "unimplemented!", "...").
Exit code 0 = all agree.  tests/test_integration_text.py runs it.
"""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RS = os.path.join(ROOT, "integration", "gpu.rs")
STRUCTS = {"BbGroupDesc": "bb_group_desc", "BbParams": "bb_params", "BbRow": "bb_row", "BbGroupInfo": "bb_group_info", "BbPolicy": "bb_policy"}
RENAME = {"type_": "type"}
PRIM = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8, "isize": 8}


def rust_type(t):
    """-> (size, align) of a Rust field type under repr(C) on x86-64"""
    t = t.strip()
    if t.startswith("*const") or t.startswith("*mut"):
        return 8, 8
    m = re.fullmatch(r"\[(\w+);\s*(\d+)\]", t)
    if m:
        s = PRIM[m.group(1)]
        return s * int(m.group(2)), s
    return PRIM[t], PRIM[t]


def parse_rust(text):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct (\w+)\s*\{(.*?)\n\}", text, re.S):
        fields, off, amax = [], 0, 1
        for f in re.finditer(r"pub (\w+):\s*([^,\n]+),", m.group(2)):
            size, align = rust_type(f.group(2))
            off = (off + align - 1) // align * align
            fields.append((RENAME.get(f.group(1), f.group(1)), off, size))
            off += size
            amax = max(amax, align)
        out[m.group(1)] = (fields, (off + amax - 1) // amax * amax)
    return out


def c_layout(structs):
    """compile the headers with a table of offsetof / sizeof for the fields named in the Rust text"""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "include/barbell_amd.h"', "int main(void) {"]
    for rs, (fields, _) in structs.items():
        cs = STRUCTS[rs]
        lines.append(f'  printf("{cs} size %zu\\n", sizeof({cs}));')
        for name, _, _ in fields:
            lines.append(f'  printf("{cs} {name} %zu %zu\\n", offsetof({cs}, {name}), sizeof((({cs}*)0)->{name}));')
    lines += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-I", ROOT, "-o", exe, src])
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    lay = {}
    for l in out.splitlines():
        p = l.split()
        if p[1] == "size":
            lay.setdefault(p[0], {})["__size__"] = int(p[2])
        else:
            lay.setdefault(p[0], {})[p[1]] = (int(p[2]), int(p[3]))
    return lay


def c_field_order(header_text, cs):
    """field names of a typedef struct in declaration order (comments stripped)"""
    flat = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    body = re.search(r"typedef struct\s*\{([^{}]*)\}\s*" + cs + r"\s*;", flat).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.search(r"(\w+)\s*(?:\[\d+\])?\s*$", part.strip()).group(1))
    return names


def main():
    bad = []
    text = open(sys.argv[1] if len(sys.argv) > 1 else RS).read()
    rs = parse_rust(text)
    if set(rs) != set(STRUCTS):
        bad.append(f"repr(C) structs in gpu.rs {sorted(rs)} != expected {sorted(STRUCTS)}")
    lay = c_layout(rs)
    hdr = open(os.path.join(ROOT, "include", "barbell_amd.h")).read() + open(os.path.join(ROOT, "include", "barbell_amd_policy.h")).read()
    from barbell_amd import _abi

    ct = {"BbGroupDesc": _abi.GroupDesc, "BbParams": _abi.Params, "BbPolicy": _abi.Policy, "BbGroupInfo": _abi.GroupInfo}
    for name, (fields, size) in rs.items():
        cs = STRUCTS[name]
        if [f[0] for f in fields] != c_field_order(hdr, cs):
            bad.append(f"{name}: fields {[f[0] for f in fields]} != {cs}'s {c_field_order(hdr, cs)}")
        if size != lay[cs]["__size__"]:
            bad.append(f"{name}: size {size} != sizeof({cs}) {lay[cs]['__size__']}")
        for fname, off, fsize in fields:
            if lay[cs].get(fname) != (off, fsize):
                bad.append(f"{name}.{fname}: offset/size {(off, fsize)} != C {lay[cs].get(fname)}")
        if name in ct:
            cst = ct[name]
            if C.sizeof(cst) != size:
                bad.append(f"{name}: size {size} != ctypes {C.sizeof(cst)}")
            for (fname, off, fsize), (cn, _) in zip(fields, cst._fields_):
                d = getattr(cst, cn)
                if (d.offset, d.size) != (off, fsize) or cn != fname:
                    bad.append(f"{name}.{fname}: {(off, fsize)} != ctypes {cn} {(d.offset, d.size)}")
    if "BbRow" in rs:
        dt = _abi.ROW_DTYPE
        for fname, off, fsize in rs["BbRow"][0]:
            if dt.fields[fname][1] != off or dt.fields[fname][0].itemsize != fsize:
                bad.append(f"BbRow.{fname}: {(off, fsize)} != numpy ROW_DTYPE {(dt.fields[fname][1], dt.fields[fname][0].itemsize)}")
    # extern block against the header's prototypes
    ext = re.search(r'extern "C" \{(.*?)\n\}', text, re.S).group(1)
    protos = {}
    flat = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for m in re.finditer(r"([\w\*\s]+?)\b(bb_\w+)\s*\(([^;{]*?)\)\s*;", flat):
        args = [a for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        protos[m.group(2)] = (len(args), m.group(1).strip())
    for m in re.finditer(r"fn (bb_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", ext, re.S):
        fn, args, ret = m.group(1), [a for a in m.group(2).split(",") if a.strip()], (m.group(3) or "").strip()
        if fn not in protos:
            bad.append(f"extern fn {fn}: not declared in include/barbell_amd.h")
            continue
        n, cret = protos[fn]
        if n != len(args):
            bad.append(f"extern fn {fn}: {len(args)} parameters, the header has {n}")
        want = {"int": "i32", "void": "", "uint32_t": "u32", "const char*": "*const c_char", "const char *": "*const c_char"}.get(cret.replace("  ", " "), None)
        if want is not None and want != ret:
            bad.append(f"extern fn {fn}: returns '{ret}', the header '{cret}'")
    for cname, val in re.findall(r"pub const (BB_\w+): \w+ = (-?\d+);", text):
        m = re.search(r"#define\s+" + cname + r"\s+\(?(-?\d+)\)?", hdr)
        if not m or int(m.group(1)) != int(val):
            bad.append(f"const {cname} = {val}, header: {m.group(1) if m else 'missing'}")
    for f in sorted(os.listdir(os.path.join(ROOT, "integration"))):
        t = open(os.path.join(ROOT, "integration", f)).read()
        for marker in ("/* retry */", "/* distinct", "todo!(", "unimplemented!(", "/* ... */"):
            if marker in t:
                bad.append(f"integration/{f}: pseudo-code marker {marker!r}")
    if bad:
        print("\n".join(bad))
        return 1
    print(f"integration/gpu.rs: {len(rs)} repr(C) structs, {len(re.findall(r'fn bb_', ext))} extern functions and the constants agree with include/*.h and barbell_amd/_abi.py")
    return 0


if __name__ == "__main__":
    sys.exit(main())
