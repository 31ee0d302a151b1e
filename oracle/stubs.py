"""TEST INFRASTRUCTURE ONLY -- documented restatements of two third-party packages
that HiTE imports on the hot path but that are absent from this image.

* ``fuzzysearch.find_near_matches`` (unpinned in /root/reference/environment.yml:29;
  call sites /root/reference/module/Util.py:9173-9174, 9511-9512, 9650, 9852-9853)
* ``Levenshtein.distance`` (python-Levenshtein, unpinned, environment.yml:44; call
  site Util.py:9403)

PARITY UNPINNED at this boundary: the real fuzzysearch package is not available, so the
definition below is what every golden fixture was generated with:

    matches  = every substring seq[s:e] with unit-cost edit distance <= max_l_dist
               to the pattern
    groups   = connected components of interval overlap (sorted by start, a match
               joins the open group while match.start < group.end)
    result   = per group the match minimising (dist, -(e-s), s), groups in start order

The real package seeds with n-grams and can pick a different start/end when an edit
falls on the first or last base of the pattern; fixtures avoid relying on that case.
"""
from collections import namedtuple

Match = namedtuple("Match", ["start", "end", "dist", "matched"])


def levenshtein(a, b):
    """Plain unit-cost edit distance (insert / delete / substitute)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def find_near_matches(subsequence, sequence, max_l_dist=0, **_ignored):
    m = len(subsequence)
    n = len(sequence)
    k = max_l_dist
    if m == 0:
        raise ValueError("Given subsequence is empty!")
    found = []
    for s in range(n):
        # DP of pattern (rows) against text[s : s+m+k] (cols); last row gives the
        # distance of the pattern to every prefix of that text window.
        w = min(m + k, n - s)
        if w < m - k or w <= 0:
            continue
        prev = list(range(w + 1))  # pattern prefix length 0
        for i in range(1, m + 1):
            cur = [i] + [0] * w
            pc = subsequence[i - 1]
            for j in range(1, w + 1):
                c = prev[j - 1] + (pc != sequence[s + j - 1])
                d = prev[j] + 1
                e = cur[j - 1] + 1
                cur[j] = min(c, d, e)
            prev = cur
        for L in range(max(1, m - k), w + 1):
            if prev[L] <= k:
                found.append((s, s + L, prev[L]))
    found.sort()
    out = []
    group = []
    gend = -1
    for mt in found:
        if group and mt[0] < gend:
            group.append(mt)
            gend = max(gend, mt[1])
        else:
            if group:
                out.append(min(group, key=lambda t: (t[2], -(t[1] - t[0]), t[0])))
            group = [mt]
            gend = mt[1]
    if group:
        out.append(min(group, key=lambda t: (t[2], -(t[1] - t[0]), t[0])))
    return [Match(s, e, d, sequence[s:e]) for (s, e, d) in out]
