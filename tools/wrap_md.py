#!/usr/bin/env python3
"""wraps the prose of a markdown file at `width` columns: table rows, headings, code fences and lines that already fit are left alone;
a list item's continuation lines keep its indentation.  usage: wrap_md.py FILE [width]"""
import re
import sys
import textwrap


def wrap(text, width=140):
    out, fence = [], False
    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
        if fence or len(line) <= width or line.lstrip().startswith("|") or line.startswith("#"):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[-*]|\d+\.)\s+)?", line)
        lead, bullet = m.group(1), m.group(2) or ""
        body = line[len(lead) + len(bullet):]
        # (two spaces after a full stop are this repo's style: keep them by protecting them through the wrap)
        body = body.replace(".  ", ".\x00 ")
        lines = textwrap.wrap(body, width=width - len(lead) - len(bullet), break_long_words=False, break_on_hyphens=False)
        cont = lead + " " * len(bullet)
        for k, l in enumerate(lines):
            out.append(((lead + bullet) if k == 0 else cont) + l.replace(".\x00 ", ".  ").replace(".\x00", "."))
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    src = open(path).read()
    open(path, "w").write(wrap(src, width))
