#!/usr/bin/env python3
"""re-flows the prose of a markdown file at `width` columns.  Table rows, headings, code fences and blank lines are kept; a paragraph or list
item (its continuation lines included) is joined and wrapped again, continuation lines indented under the item's text.  A line that opens
with emphasis (`**Bold.**`, `*Italic*`) starts a new paragraph of its own, as this repository writes them.  usage: wrap_md.py FILE [width]"""
import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)((?:[-*]|\d+[a-z]?\.)\s+)")


def flush(block, out, width):
    if not block:
        return
    first = block[0]
    m = ITEM.match(first)
    lead, bullet = (m.group(1), m.group(2)) if m else (re.match(r"^\s*", first).group(0), "")
    body = " ".join([first[len(lead) + len(bullet):].strip()] + [l.strip() for l in block[1:]])
    body = body.replace(".  ", ".\x00 ")    # (two spaces after a full stop are this repository's style: protected through the wrap)
    lines = textwrap.wrap(body, width=width - len(lead) - len(bullet), break_long_words=False, break_on_hyphens=False) or [""]
    cont = lead + " " * len(bullet)
    for k, l in enumerate(lines):
        out.append(((lead + bullet) if k == 0 else cont) + l.replace(".\x00 ", ".  ").replace(".\x00", "."))
    block.clear()


def wrap(text, width=140):
    out, block, fence = [], [], False
    for line in text.split("\n"):
        stripped = line.lstrip()
        if stripped.startswith("```"):
            flush(block, out, width)
            fence = not fence
            out.append(line)
            continue
        if fence or not stripped or stripped.startswith("|") or line.startswith("#") or stripped.startswith("<"):
            flush(block, out, width)
            out.append(line)
            continue
        if ITEM.match(line) or re.match(r"^\*{1,2}[A-Za-z`(]", stripped):
            flush(block, out, width)
        block.append(line)
    flush(block, out, width)
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    src = open(path).read()
    open(path, "w").write(wrap(src, width))
