"""Hygiene helper: re-flow over-long COMMENTS of the C++ / HIP sources to <= WIDTH columns (code is never re-flowed).
  * a line that is only a `//` comment is split at spaces into several `//` lines of the same indent;
  * a code line with a trailing `//` comment that makes it too long gets the comment moved to `//` lines of its own ABOVE it.
Lines of macros (ending in a backslash, or following one), preprocessor lines and lines whose `//` sits inside a string literal are left alone.
usage: python scripts/wrap_comments.py [--width 160] [--code] file...     (--code: over-long code lines are broken at commas inside parentheses as well)"""
import re
import sys

WIDTH = 160
RULER = re.compile(r"[-=*#]{8,}")


def split_comment(indent, text, lead=""):
    words, out, cur = [w for w in text.split(" ") if w != ""] if lead else text.split(" "), [], ""
    indent_c = indent + "// " + lead
    room = max(40, WIDTH - len(indent_c))
    for w in words:
        if cur and len(cur) + 1 + len(w) > room:
            out.append((indent_c if not out or not lead else indent_c + "  ") + cur)
            cur = w
        else:
            cur = w if not cur else cur + " " + w
    if cur:
        out.append((indent_c if not out or not lead else indent_c + "  ") + cur)
    return out


def find_trailing_comment(line):
    """index of the `//` that starts a trailing comment, or -1 (skips string / char literals)"""
    i, n, q = 0, len(line), None
    while i < n - 1:
        c = line[i]
        if q:
            if c == "\\":
                i += 2
                continue
            if c == q:
                q = None
        elif c in "\"'":
            q = c
        elif c == "/" and line[i + 1] == "/":
            return i
        elif c == "/" and line[i + 1] == "*":
            return -1
        i += 1
    return -1


def reflow_paragraphs(src):
    """consecutive pure-comment lines of one indent whose text starts right after `// ` form a paragraph: if one of its lines is too long the whole paragraph is re-flowed
    (a line that starts with more spaces -- a diagram, a table row, an indented list -- or with a list marker ends the paragraph and is never joined)"""
    out, i, n, changed = [], 0, len(src), 0
    pat = re.compile(r"^(\s*)// ?(\S.*)$")
    while i < n:
        m = pat.match(src[i])
        if not m or src[i].rstrip().endswith("\\") or re.match(r"^\s*// {2,}", src[i]) or RULER.search(src[i]):
            out.append(src[i])
            i += 1
            continue
        indent, para, j = m.group(1), [m.group(2)], i + 1
        while j < n:
            mj = pat.match(src[j])
            if not mj or mj.group(1) != indent or RULER.search(src[j]) or re.match(r"^\s*// {2,}", src[j]) or re.match(r"^(\(?[0-9a-z]\)|[-*#]|[0-9]+\.)\s", mj.group(2)) or src[j].rstrip().endswith("\\"):
                break
            para.append(mj.group(2))
            j += 1
        if any(len(l) > WIDTH for l in src[i:j]) and not any(re.match(r"^[-=*#]{8,}", b) for b in para):
            out.extend(split_comment(indent, " ".join(b.strip() for b in para)))
            changed += 1
        else:
            out.extend(src[i:j])
        i = j
    return out, changed


def process(path):
    src = open(path).read().split("\n")
    src, changed = reflow_paragraphs(src)
    out, in_macro = [], False
    for line in src:
        cont = line.rstrip().endswith("\\")
        if len(line) <= WIDTH or in_macro or cont or line.lstrip().startswith("#") or "\t" in line:
            out.append(line)
            in_macro = cont
            continue
        in_macro = cont
        m = re.match(r"^(\s*)//( ?)(.*)$", line)
        if m:
            body = m.group(3)
            if RULER.search(body):  # rulers
                out.append(line[:WIDTH].rstrip())
            else:
                out.extend(split_comment(m.group(1), body.lstrip(), " " * (len(body) - len(body.lstrip()))))
            changed += 1
            continue
        k = find_trailing_comment(line)
        if k > 0 and line[:k].strip():
            code, com = line[:k].rstrip(), line[k + 2:].strip()
            indent = re.match(r"^(\s*)", line).group(1)
            if code.strip() in ("}", "{", "};") or code.rstrip().endswith("else"):
                out.append(line)
                continue
            out.extend(split_comment(indent, com))
            out.append(code)
            changed += 1
            continue
        out.append(line)
    if changed:
        open(path, "w").write("\n".join(out))
    left = sum(1 for l in out if len(l) > WIDTH)
    print("%s: %d comments re-flowed, %d lines still > %d (code)" % (path, changed, left, WIDTH))


def wrap_code_line(line):
    """break an over-long CODE line at a `, ` that lies inside parentheses and outside string / char literals; the rest goes to continuation lines aligned one past the
    innermost open parenthesis that is still open at the break (capped).  Lines of macros and preprocessor lines are left alone by the caller."""
    out = []
    while len(line) > WIDTH:
        depth, q, opens, best, i, n = 0, None, [], -1, 0, len(line)
        best_open = None
        while i < min(n, WIDTH):
            c = line[i]
            if q:
                if c == "\\":
                    i += 2
                    continue
                if c == q:
                    q = None
            elif c in "\"'":
                q = c
            elif c == "/" and i + 1 < n and line[i + 1] == "/":
                break
            elif c in "([{":
                depth += 1
                opens.append(i)
            elif c in ")]}":
                depth -= 1
                if opens:
                    opens.pop()
            elif c == "," and depth >= 1 and i + 1 < n and line[i + 1] == " " and i < WIDTH - 1:
                best, best_open = i, (opens[-1] if opens else None)
            i += 1
        if best < 0:
            break
        indent = len(line) - len(line.lstrip())
        col = min((best_open + 1) if best_open is not None else indent + 4, indent + 24)
        col = max(col, indent + 2)
        out.append(line[:best + 1])
        line = " " * col + line[best + 2:]
    out.append(line)
    return out


def process_code(path):
    src = open(path).read().split("\n")
    out, changed, in_macro = [], 0, False
    for line in src:
        cont = line.rstrip().endswith("\\")
        if len(line) <= WIDTH or in_macro or cont or line.lstrip().startswith("#") or line.lstrip().startswith("//") or "\t" in line:
            out.append(line)
            in_macro = cont
            continue
        in_macro = cont
        pieces = wrap_code_line(line)
        changed += len(pieces) > 1
        out.extend(pieces)
    if changed:
        open(path, "w").write("\n".join(out))
    print("%s: %d code lines wrapped, %d still > %d" % (path, changed, sum(1 for l in out if len(l) > WIDTH), WIDTH))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--width":
        WIDTH = int(args[1])
        args = args[2:]
    code = bool(args) and args[0] == "--code"
    if code:
        args = args[1:]
    for p in args:
        process(p)
        if code:
            process_code(p)
