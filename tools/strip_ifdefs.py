"""Remove preprocessor branches of KNOWN symbols from a source file (a small unifdef): experiment switches that were measured
and decided leave the product source this way; the removed variants live on as patches under tools/experiments/.
  python tools/strip_ifdefs.py file.hip SYM=0 SYM2=undef SYM3=1 ... > out
Handled forms: #ifdef S, #ifndef S, #if S, #if !S, #if S > 0, #if S == N, #elif defined(S), #else, #endif (nesting kept for
conditions on other symbols).  `#ifndef S / #define S v / #endif` default blocks of a stripped symbol are removed as well."""
import re, sys

def main():
    path, syms = sys.argv[1], {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        syms[k] = None if v == "undef" else int(v)
    lines = open(path).read().split("\n")
    out, stack = [], []          # stack entries: [known, taken_any, active_now, parent_active]

    def active():
        return all(e[2] for e in stack if e[0]) and all(e[3] for e in stack)

    def ev(cond):
        """-> True / False / None (unknown)"""
        c = cond.strip()
        m = re.fullmatch(r"defined\s*\(?\s*(\w+)\s*\)?", c)
        if m and m.group(1) in syms: return syms[m.group(1)] is not None
        m = re.fullmatch(r"!\s*defined\s*\(?\s*(\w+)\s*\)?", c)
        if m and m.group(1) in syms: return syms[m.group(1)] is None
        m = re.fullmatch(r"(\w+)", c)
        if m and m.group(1) in syms: return bool(syms[m.group(1)] or 0)
        m = re.fullmatch(r"!\s*(\w+)", c)
        if m and m.group(1) in syms: return not bool(syms[m.group(1)] or 0)
        m = re.fullmatch(r"(\w+)\s*(>|==|>=|<|!=)\s*(\d+)", c)
        if m and m.group(1) in syms:
            v, n = syms[m.group(1)] or 0, int(m.group(3))
            return {">": v > n, "==": v == n, ">=": v >= n, "<": v < n, "!=": v != n}[m.group(2)]
        return None

    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        # default-definition block of a stripped symbol
        m = re.fullmatch(r"#ifndef\s+(\w+)", s)
        if m and m.group(1) in syms and i + 2 < len(lines) and re.match(r"#define\s+" + m.group(1) + r"\b", lines[i + 1].strip()) \
                and lines[i + 2].strip().startswith("#endif"):
            i += 3
            continue
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if active(): out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), re.sub(r"//.*", "", m.group(2)).strip()
        if kind in ("ifdef", "ifndef", "if"):
            cond = rest if kind == "if" else ("defined(%s)" % rest if kind == "ifdef" else "!defined(%s)" % rest)
            r = ev(cond)
            par = active()
            if r is None:
                stack.append([False, False, True, par])
                if par: out.append(ln)
            else:
                stack.append([True, r, r, par])
        elif kind == "elif":
            e = stack[-1]
            if not e[0]:
                if active(): out.append(ln)
            else:
                r = ev(rest)
                assert r is not None, "elif on an unknown condition after a known one: " + ln
                e[2] = (not e[1]) and r
                e[1] = e[1] or r
        elif kind == "else":
            e = stack[-1]
            if not e[0]:
                if active(): out.append(ln)
            else:
                e[2] = not e[1]
                e[1] = True
        else:
            e = stack.pop()
            if not e[0] and e[3] and all(x[2] for x in stack if x[0]) and all(x[3] for x in stack): out.append(ln)
        i += 1
    assert not stack
    sys.stdout.write("\n".join(out))

main()
