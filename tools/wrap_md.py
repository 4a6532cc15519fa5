#!/usr/bin/env python3
"""Re-wraps the prose of a Markdown file at a readable width (default 118 columns): paragraphs and list items are re-flowed, headings, tables, code fences and
indented code stay as they are. usage: tools/wrap_md.py <file.md> [width]"""
import re, sys, textwrap
path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path).read().split("\n")
out = []; para = []; fence = False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([*\-+]|\d+\.|\([a-z0-9]+\))\s+", first)
    if m:
        indent = " " * len(m.group(0)); head = m.group(0)
        text = first[len(head):] + " " + " ".join(l.strip() for l in para[1:])
        out.extend(textwrap.wrap(text.strip(), width=width, initial_indent=head, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    else:
        lead = re.match(r"^\s*", first).group(0)
        text = " ".join(l.strip() for l in para)
        out.extend(textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=lead, break_long_words=False, break_on_hyphens=False))
    para = []


for ln in lines:
    if ln.strip().startswith("```"):
        flush(); fence = not fence; out.append(ln); continue
    if fence or ln.startswith("    ") and not para:
        out.append(ln); continue
    if not ln.strip():
        flush(); out.append(ln); continue
    if ln.lstrip().startswith(("|", "#", ">")) or re.match(r"^\s*[-=]{3,}\s*$", ln):
        flush(); out.append(ln); continue
    if re.match(r"^\s*([*\-+]|\d+\.)\s+", ln) and para:
        flush()
    para.append(ln)
flush()
open(path, "w").write("\n".join(out))
