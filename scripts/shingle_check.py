#!/usr/bin/env python3
"""Identifier-normalised token-shingle overlap of the functions of one of our C files with a span of a reference file.

    scripts/shingle_check.py damar_amd/csrc/host/redundancy.c /root/reference/dalign/filter.c 1573 2077 [n=12]

For every top-level function of OURS: the fraction of its n-token shingles (identifiers -> ID, numbers -> NUM,
comments and strings dropped) that also occur in the reference span.  Development aid for the copy check
(no function of host/redundancy.c above 20 % at n = 12); reads the reference as text only.
"""
import re, sys

KEYWORDS = set("if else for while do return continue break int static const void struct typedef enum sizeof "
               "unsigned char long short double float switch case default goto".split())
TOK = re.compile(r'[A-Za-z_]\w*|\d+\.?\d*|\.\d+|->|<<|>>|<=|>=|==|!=|&&|\|\||\+\+|--|[-+*/%&|^!~<>=?:;,.(){}\[\]]')

def strip(text):
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    text = re.sub(r'"(\\.|[^"\\])*"', ' STR ', text)
    text = re.sub(r"'(\\.|[^'\\])'", ' CHR ', text)
    text = re.sub(r'^\s*#.*$', ' ', text, flags=re.M)
    return text

def norm(text):
    out = []
    for t in TOK.findall(strip(text)):
        if t[0].isalpha() or t[0] == '_':
            out.append(t if t in KEYWORDS else 'ID')
        elif t[0].isdigit() or (t[0] == '.' and len(t) > 1):
            out.append('NUM')
        else:
            out.append(t)
    return out

def shingles(toks, n):
    return {tuple(toks[i:i + n]) for i in range(len(toks) - n + 1)}

def functions(text):
    """(name, body text) of top-level brace blocks that follow a ')'."""
    s = strip(text)
    depth, start, res, i = 0, None, [], 0
    last_close = -1
    while i < len(s):
        c = s[i]
        if c == '{':
            if depth == 0:
                head = s[max(0, last_close + 1):i]
                if head.rstrip().endswith(')'):
                    start = (i, head)
            depth += 1
        elif c == '}':
            depth -= 1
            if depth == 0:
                if start is not None:
                    m = re.findall(r'([A-Za-z_]\w*)\s*\(', start[1])
                    res.append((m[0] if m else '?', start[1] + s[start[0]:i + 1]))
                start = None
                last_close = i
        elif c == ';' and depth == 0:
            last_close = i
        i += 1
    return res

def main():
    ours, ref, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 12
    ref_lines = open(ref, errors='replace').read().split('\n')[lo - 1:hi]
    R = shingles(norm('\n'.join(ref_lines)), n)
    worst = 0.0
    for name, body in functions(open(ours).read()):
        S = shingles(norm(body), n)
        if not S:
            continue
        f = len(S & R) / len(S)
        worst = max(worst, f)
        print(f"{name:32s} {len(S):5d} shingles  {100 * f:5.1f} %")
    print(f"worst {100 * worst:.1f} % (n = {n})")

if __name__ == '__main__':
    main()
