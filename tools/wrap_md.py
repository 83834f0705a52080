#!/usr/bin/env python3
"""Re-wraps the prose of a Markdown file at 120 columns: paragraphs and list items are re-flowed, tables, fenced code, headings and
indented code are left alone.  usage: wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, para, fence = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", first)
    if m:
        init = m.group(0)
        sub = " " * len(init)
        body = first[len(init):] + " " + " ".join(s.strip() for s in para[1:])
    else:
        ind = re.match(r"^\s*", first).group(0)
        init = sub = ind
        body = " ".join(s.strip() for s in para)
    out.extend(textwrap.wrap(body.strip(), width=width, initial_indent=init, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False) or [init.rstrip()])
    para = []


for line in open(path).read().split("\n"):
    s = line.rstrip()
    if s.lstrip().startswith("```"):
        flush(); fence = not fence; out.append(s); continue
    if fence or s.lstrip().startswith("|") or s.startswith("#") or not s.strip():
        flush(); out.append(s); continue
    if re.match(r"^\s*([*+-]|\d+\.)\s+", s):
        flush(); para = [s]; continue
    para.append(s)
flush()
open(path, "w").write("\n".join(out).rstrip("\n") + "\n")
