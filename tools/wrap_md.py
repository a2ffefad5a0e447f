#!/usr/bin/env python3
"""Re-wrap a markdown file at WIDTH columns without changing a word: paragraphs and list items are re-flowed (hanging indent kept), code fences, headings and
tables are left alone -- except tables with a row longer than WIDTH, which become bullet lists ("**first cell** -- header: cell; header: cell ...": a 600-column
table row is not a table any more).  usage: tools/wrap_md.py FILE [WIDTH=160]   (round 6: the review's hygiene item for DESIGN.md / README.md)"""
import re
import sys
import textwrap

path, width = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 160
lines = open(path).read().split("\n")
out, i, n = [], 0, len(lines)
ITEM = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")


def flow(text, first, rest):
    w = textwrap.TextWrapper(width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)
    return w.wrap(text) or [first.rstrip()]


def cells(row):
    body = row.strip()
    body = body[1:] if body.startswith("|") else body
    body = body[:-1] if body.endswith("|") else body
    return [c.strip() for c in re.split(r"(?<!\\)\|", body)]


while i < n:
    ln = lines[i]
    if ln.strip().startswith("```"):                       # code fence: verbatim
        out.append(ln); i += 1
        while i < n and not lines[i].strip().startswith("```"):
            out.append(lines[i]); i += 1
        if i < n:
            out.append(lines[i]); i += 1
        continue
    if ln.lstrip().startswith("|"):                        # table block
        j = i
        while j < n and lines[j].lstrip().startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(b) for b in block) <= width or len(block) < 3:
            out.extend(block)
        else:
            hdr = cells(block[0])
            for row in block[2:]:
                c = cells(row)
                parts = [f"{h}: {v}" if h else v for h, v in zip(hdr[1:], c[1:]) if v]
                text = (c[0] + " -- " if c[0] else "") + "; ".join(parts)
                out.extend(flow(text, "* ", "  "))
        i = j
        continue
    if not ln.strip() or ln.startswith("#") or ln.startswith("<") or re.match(r"^\s*(---+|===+)\s*$", ln):
        out.append(ln); i += 1
        continue
    m = ITEM.match(ln)
    indent = m.group(1) if m else re.match(r"^\s*", ln).group(0)
    first = (m.group(0) if m else indent)
    rest = " " * len(first) if m else indent
    buf = [ln[len(first):].strip()]
    i += 1
    while i < n:                                          # continuation lines of the same paragraph / item
        nx = lines[i]
        if (not nx.strip() or nx.startswith("#") or nx.lstrip().startswith("|") or nx.strip().startswith("```") or ITEM.match(nx) or nx.startswith("<")):
            break
        buf.append(nx.strip()); i += 1
    out.extend(flow(" ".join(buf), first, rest))
open(path, "w").write("\n".join(out))
