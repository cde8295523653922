#!/usr/bin/env python3
"""Wraps the C++ / HIP sources under rpg_open_remode_amd/csrc (and include/) to a column limit WITHOUT touching a token of code: a comment line
that is too long is re-flowed over several comment lines of the same indentation; a trailing comment that makes a code line too long moves onto
its own line(s) above the code.  Lines whose CODE is too long are listed for the hand.   usage: python tools/wrap_columns.py [--limit 140] [--check] files..."""
import argparse, re, sys, textwrap

ap = argparse.ArgumentParser()
ap.add_argument("--limit", type=int, default=140); ap.add_argument("--check", action="store_true"); ap.add_argument("files", nargs="+")
a = ap.parse_args()


def comment_start(line):
    """index of the // that starts a trailing comment (outside string / char literals), or -1"""
    in_s, q, i = False, "", 0
    while i < len(line) - 1:
        c = line[i]
        if in_s:
            if c == "\\":
                i += 2
                continue
            if c == q:
                in_s = False
        elif c in "\"'":
            in_s, q = True, c
        elif c == "/" and line[i + 1] == "/":
            return i
        i += 1
    return -1


def flow(indent, text, limit):
    body = text.strip()
    width = max(40, limit - len(indent) - 3)
    return [f"{indent}// {t}" for t in textwrap.wrap(body, width=width, break_long_words=False, break_on_hyphens=False)] or [f"{indent}//"]


def spaces_outside_literals(code):
    """indices of the blanks of `code` that are outside string / char literals"""
    out, in_s, q, i = [], False, "", 0
    while i < len(code):
        c = code[i]
        if in_s:
            if c == "\\":
                i += 2
                continue
            if c == q:
                in_s = False
        elif c in "\"'":
            in_s, q = True, c
        elif c == " ":
            out.append(i)
        i += 1
    return out


def break_code(code, limit, indent):
    """`code` (no comment) over several lines: outside literals and directives a blank may become a line break anywhere, so no token is touched.
    Break points by preference: behind a comma, in front of && / || / ? / :, behind = , in front of an arithmetic or shift operator, any blank."""
    lines = []
    cont = indent + "    "
    while len(code) > limit:
        blanks = [i for i in spaces_outside_literals(code) if len(indent) + 8 < i <= limit]
        if not blanks:
            return None
        def score(i):
            before, after = code[:i], code[i + 1:]
            if before.endswith(","): return 5
            if before.endswith("{") and after.startswith("return"): return 6
            if after.startswith(("&& ", "|| ")): return 4
            if after.startswith(("? ", ": ")): return 3
            if before.endswith((" =", "return")): return 2 if before.endswith(" =") else 0
            if after.startswith(("+ ", "- ", "* ", "/ ", "<< ", ">> ", "| ", "& ", "^ ")): return 2
            if before.endswith("("): return 1
            return 0
        # the best kind of break point, and among those the right-most that still leaves a reasonably filled line
        best = max(blanks, key=lambda i: (score(i) if i > limit * 0.55 else score(i) - 3, i))
        lines.append(code[:best].rstrip())
        code = cont + code[best + 1:].lstrip()
    lines.append(code)
    return lines


BULLET = re.compile(r"^\s*([*\-]|\d+\.|\([a-z0-9]\)|[A-Za-z_]+\s{2,})\s*")


def reflow_comment_blocks(src, limit):
    """re-flows, paragraph by paragraph, every block of whole-line comments in which some line is too long (a paragraph: consecutive comment lines of
    one indentation up to an empty comment line, a bullet or a line that is indented further than the one before it)"""
    out, i, changed = [], 0, False
    while i < len(src):
        m = re.match(r"^(\s*)//(.*)$", src[i])
        if not m:
            out.append(src[i]); i += 1
            continue
        indent = m.group(1)
        j = i
        block = []
        while j < len(src):
            mm = re.match(r"^(\s*)//(.*)$", src[j])
            if not mm or mm.group(1) != indent:
                break
            block.append(mm.group(2))
            j += 1
        # paragraphs
        paras, cur = [], []
        for t in block:
            lead = len(t) - len(t.lstrip(" "))
            starts_new = (not t.strip()) or BULLET.match(t) is not None and lead >= 1 and (t.lstrip()[:1] in "*-" or lead >= 2) or (cur and lead > (len(cur[-1]) - len(cur[-1].lstrip(" "))) + 1)
            if starts_new and cur:
                paras.append(cur); cur = []
            if not t.strip():
                paras.append([t])
            else:
                cur.append(t)
        if cur:
            paras.append(cur)
        for para in paras:
            if all(len(indent) + 2 + len(t) <= limit for t in para) or not para[0].strip():
                out += [f"{indent}//{t}" for t in para]
                continue
            first_lead = para[0][:len(para[0]) - len(para[0].lstrip(" "))]
            hang = para[1][:len(para[1]) - len(para[1].lstrip(" "))] if len(para) > 1 else (first_lead if not BULLET.match(para[0]) else first_lead + "  ")
            text = " ".join(t.strip() for t in para)
            # keep double spaces after full stops as the sources write them
            width = limit - len(indent) - 2
            wrapped = textwrap.wrap(text, width=width, initial_indent=first_lead or " ", subsequent_indent=hang or " ", break_long_words=False, break_on_hyphens=False)
            out += [f"{indent}//{t}" for t in wrapped]
            changed = True
        i = j
    return out, changed


left = 0
for path in a.files:
    src, block_changed = reflow_comment_blocks(open(path).read().split("\n"), a.limit)
    out, changed = [], block_changed
    for n, line in enumerate(src, 1):
        if len(line) <= a.limit or line.rstrip().endswith("\\"):
            out.append(line)
            continue
        indent = re.match(r"\s*", line).group(0)
        stripped = line.strip()
        if stripped.startswith("//"):
            out += flow(indent, stripped[2:], a.limit)
            changed = True
            continue
        c = comment_start(line)
        if c > 0 and len(line[:c].rstrip()) <= a.limit:
            out += flow(indent, line[c + 2:], a.limit)
            out.append(line[:c].rstrip())
            changed = True
            continue
        code = line[:c].rstrip() if c > 0 else line.rstrip()
        broken = None if stripped.startswith("#") else break_code(code, a.limit, indent)
        if broken:
            if c > 0:
                out += flow(indent, line[c + 2:], a.limit)
            out += broken
            changed = True
            continue
        left += 1
        print(f"{path}:{n}: code of {len(code)} columns: wrap by hand")
        out.append(line)
    if changed and not a.check:
        open(path, "w").write("\n".join(out))
print(f"{left} lines left for the hand")
sys.exit(1 if (a.check and left) else 0)
