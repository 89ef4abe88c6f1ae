#!/usr/bin/env python3
"""Re-wrap the paragraphs and bullets of a markdown file to WIDTH columns (tables, headings, code fences and blank lines stay as they
are; a bullet's continuation lines are indented by two spaces).  usage: reflow_md.py FILE [WIDTH=120]"""
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, block, kind, fence = [], [], None, False


def flush():
    global block, kind
    if block:
        text = " ".join(x.strip() for x in block)
        if kind == "bullet":
            out.extend(textwrap.wrap(text, width, initial_indent="", subsequent_indent="  ", break_long_words=False, break_on_hyphens=False))
        else:
            out.extend(textwrap.wrap(text, width, break_long_words=False, break_on_hyphens=False))
    block, kind = [], None


for ln in lines:
    st = ln.strip()
    if st.startswith("```"):
        flush(); out.append(ln); fence = not fence; continue
    if fence or st == "" or st.startswith("#") or st.startswith("|"):
        flush(); out.append(ln); continue
    if st.startswith("* ") or st.startswith("- "):
        flush(); block, kind = [st], "bullet"; continue
    if kind is None:
        kind = "para"
    block.append(st)
flush()
open(path, "w").write("\n".join(out))
