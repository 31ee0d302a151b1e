"""Device functions of the HIP sources that are plain integer code are compiled for the HOST (g++) straight from the .hip file
and compared with the CPU twins on random inputs -- a CPU-side check of the kernel logic itself (no GPU needed; the GPU
parity tests then cover launch geometry and memory layout).  Blocks are cut out between `// >>> name` and `// <<< name`."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402

PRELUDE = r"""
#include <stdint.h>
#include <algorithm>
#define __device__
#define __forceinline__ inline
#define __noinline__
#define __restrict__
using std::min;
using std::max;
"""


def _block(path, name):
    src = open(path).read()
    m = re.search(r"// >>> %s.*?\n(.*?)// <<< %s" % (name, name), src, re.S)
    assert m, name
    return m.group(1)


def _build(tmp_path, name, body, wrapper):
    cpp = tmp_path / (name + ".cpp")
    so = tmp_path / (name + ".so")
    cpp.write_text(PRELUDE + body + wrapper)
    extra = os.environ.get("HITE_HOST_CXXFLAGS", "").split()       # e.g. -fsanitize=address,undefined (with libasan preloaded)
    subprocess.run(["g++", "-O2", "-shared", "-fPIC"] + extra + ["-o", str(so), str(cpp)], check=True)
    return C.CDLL(str(so))


def _ext_lib(tmp_path):
    body = _block(os.path.join(ROOT, "hite_amd", "csrc", "hite_ext.h"), "ext_align_dev")
    # (on the device the column-minimum tables of the bit-parallel band live in LDS; here they are a plain array)
    return _build(tmp_path, "ext", "#define EXT_LUT_PTR const uint32_t *\n" + body, r"""
static uint32_t g_lut[EXT_LUT_WORDS];
static bool g_lut_ready = false;
static const uint32_t *host_lut() {
    if (!g_lut_ready) { for (int t = 0; t < EXT_LUT_WORDS; t++) g_lut[t] = ext_lut_entry(t >> 8, t & 255); g_lut_ready = true; }
    return g_lut;
}
extern "C" void host_ext(const uint8_t *q, int64_t p0, int step, int comp, int n, const uint32_t *bases, const uint32_t *nmask,
                         int64_t g0, int dir, int64_t jmax, int *i_out, int *t_out) {
    int s;
    ext_align_dev<ExtCopyMode>(q, p0, step, comp != 0, n, bases, nmask, g0, dir, jmax, -EXT_B, EXT_B, i_out, t_out, &s, host_lut());
}
// the bit-parallel band against the cell-by-cell band, column by column: both states walk the same extension; after every column
// the 17 cell values (ext_expand of a copy of the fast state), the best score so far and its position must agree.
// returns -1, or the first column at which they differ
extern "C" int host_ext_lockstep(const uint8_t *q, int64_t p0, int step, int comp, int n, const uint32_t *bases, const uint32_t *nmask,
                                 int64_t g0, int dir, int64_t jmax, int *n_fast) {
    ExtStateT<ExtCopyMode> A, B;
    ext_init(A, q, p0, step, comp != 0, n, bases, nmask, g0, dir, jmax);
    ext_init(B, q, p0, step, comp != 0, n, bases, nmask, g0, dir, jmax);
    B.fast = false;
    *n_fast = 0;
    if (n < 1) return -1;
    for (;;) {
        const bool was_fast = A.fast;
        const bool da = ext_step(A, bases, nmask, host_lut()), db = ext_step(B, bases, nmask, host_lut());
        *n_fast += was_fast && A.fast;
        if (da != db || A.best_i != B.best_i || A.best_t != B.best_t || A.best_s != B.best_s || A.i != B.i) return B.i;
        if (da) return -1;
        ExtStateT<ExtCopyMode> X = A;
        if (X.fast) ext_expand(X);
        for (int b = 0; b < EXT_W; b++) {
            const int va = X.D[b] >= EXT_INF ? EXT_INF : X.D[b], vb = B.D[b] >= EXT_INF ? EXT_INF : B.D[b];
            if (va != vb) return B.i;
        }
    }
}
extern "C" void host_ext_tandem(int64_t p0, int step, int n, const uint32_t *bases, const uint32_t *nmask,
                                int64_t g0, int dir, int64_t jmax, int dlo, int dhi, int *i_out, int *t_out, int *s_out) {
    ext_align_dev<ExtTandemMode>(nullptr, p0, step, false, n, bases, nmask, g0, dir, jmax, dlo, dhi, i_out, t_out, s_out, nullptr);
}
""")


def _pack(genome):
    G = len(genome)
    codes = np.zeros(G + 64, np.uint32)
    lut = np.full(256, 0, np.uint32)
    lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
    codes[:G] = lut[genome]
    bases = np.zeros((G + 64) // 16 + 2, np.uint32)
    for k in range(16):
        part = codes[k::16]
        bases[:len(part)] |= part << np.uint32(2 * k)
    nm = np.zeros((G + 64) // 32 + 2, np.uint32)
    isn = np.zeros(G + 64, np.uint32)
    isn[:G] = ~np.isin(genome, [65, 67, 71, 84])
    for k in range(32):
        part = isn[k::32]
        nm[:len(part)] |= part << np.uint32(k)
    return bases, nm


def test_ext_align_tandem_mode_vs_twin(tmp_path):
    """the same device function in the tandem-repeat masker's mode (query = the packed genome itself, match 2 / edit 5 as
    S = 2 i - 7 cost, diagonals that pair a base with itself excluded) == orc_ext_align_scored"""
    lib = _ext_lib(tmp_path)
    L = O.lib()
    L.orc_ext_align_scored.restype = C.c_int64
    rng = np.random.default_rng(8)
    n_pos = 0
    for case in range(300):
        G = int(rng.integers(200, 3000))
        genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=G)
        p = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 12, 20, 33, 77, 150]))
        a0 = int(rng.integers(20, G // 2))
        copies = float(rng.choice([1.5, 2, 3, 6, 12]))
        Lr = min(int(p * copies) + int(rng.integers(0, 30)), G - a0 - 10)
        unit = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=p)
        arr = []
        div, ind = float(rng.choice([0, 0.03, 0.1])), float(rng.choice([0, 0.01, 0.03]))
        while len(arr) < Lr:
            for ch in unit:
                x = rng.random()
                if x < ind / 2:
                    continue
                if x < ind:
                    arr.append(int(rng.choice([65, 67, 71, 84])))
                arr.append(int(ch) if rng.random() >= div else int(rng.choice([65, 67, 71, 84])))
        genome[a0:a0 + Lr] = arr[:Lr]
        if case % 9 == 0:
            genome[rng.integers(0, G, size=2)] = ord("N")
        bases, nm = _pack(genome)
        s = a0 + int(rng.integers(0, max(1, Lr // 2)))
        cb, ce = (0, G) if case % 4 else (max(0, s - int(rng.integers(0, 60))), min(G, s + p + int(rng.integers(0, 200))))
        lim = min(p - 1, 8)
        for d in (+1, -1):
            if d > 0:
                n = max(0, min(ce - (s + p), 4096))
                seg = genome[s:s + n].copy()
                dlo, dhi = -lim, 8
            else:
                n = max(0, min(s - cb, 4096))
                seg = genome[s - n:s][::-1].copy()
                dlo, dhi = -8, lim
            if s + p > ce:
                continue
            t_ref, s_ref = C.c_int64(0), C.c_int64(0)
            gbuf = np.ascontiguousarray(genome)
            segb = np.ascontiguousarray(np.concatenate([seg, np.zeros(1, np.uint8)]))
            i_ref = L.orc_ext_align_scored(segb.ctypes.data_as(O.u8p), C.c_int64(n), d, gbuf.ctypes.data_as(O.u8p), C.c_int64(s + p), C.c_int64(cb),
                                           C.c_int64(ce), 2, 7, 30, dlo, dhi, C.byref(t_ref), C.byref(s_ref))
            jmax = (ce - (s + p)) if d > 0 else (s + p - cb)
            io, to, so = C.c_int(0), C.c_int(0), C.c_int(0)
            lib.host_ext_tandem(C.c_int64(s if d > 0 else s - 1), d, n, bases.ctypes.data_as(C.POINTER(C.c_uint32)),
                                nm.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(s + p), d, C.c_int64(jmax), dlo, dhi, C.byref(io), C.byref(to), C.byref(so))
            assert (io.value, to.value, so.value) == (int(i_ref), int(t_ref.value), int(s_ref.value)), (case, p, d, s, n, io.value, to.value, so.value, i_ref, t_ref.value, s_ref.value)
            n_pos += i_ref > 20
    assert n_pos > 100


def test_ext_align_device_function_vs_twin(tmp_path):
    """chain end extension (hite_copies.hip:ext_align_dev) == oracle/hite_oracle_copies.c:ext_align: aligned bases and genome
    bases used, both directions, both strands, contig borders inside the band, N bases, unrelated sequence (x-drop)"""
    lib = _ext_lib(tmp_path)
    L = O.lib()
    L.orc_ext_align.restype = C.c_int64
    rng = np.random.default_rng(7)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    n_cases = n_cut = n_full = n_fast_cols = 0
    for case in range(1600):
        G = int(rng.integers(80, 900)) if case % 8 else int(rng.integers(2500, 5000))
        genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=G)
        if case % 7 == 0:
            genome[rng.integers(0, G, size=3)] = ord("N")
        # packed form the kernels read (16 bases per word + 1-bit mask), with slack words
        codes = np.zeros(G + 64, np.uint32)
        lut = np.full(256, 0, np.uint32)
        lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
        codes[:G] = lut[genome]
        bases = np.zeros((G + 64) // 16 + 2, np.uint32)
        for k in range(16):
            part = codes[k::16]
            bases[:len(part)] |= part << np.uint32(2 * k)
        nm = np.zeros((G + 64) // 32 + 2, np.uint32)
        isn = np.zeros(G + 64, np.uint32)
        isn[:G] = (genome == ord("N"))
        for k in range(32):
            part = isn[k::32]
            nm[:len(part)] |= part << np.uint32(k)
        d = +1 if case % 2 == 0 else -1
        n = int(rng.integers(0, min(G - 20, 400 if case % 8 else 2048)))
        g0 = int(rng.integers(10, G - 10))
        gmin, gmax = (0, G) if case % 5 else (int(rng.integers(0, g0 + 1)), int(rng.integers(g0, G + 1)))
        # query segment in walking order: a diverged copy of the genome it walks over for `hom` bases, then unrelated bases
        hom = int(rng.integers(0, n + 1)) if case % 3 else n
        walk = genome[g0:g0 + n + 40] if d > 0 else genome[max(0, g0 - n - 40):g0][::-1]
        seg = []
        wi = 0
        div = float(rng.choice([0.0, 0.05, 0.15, 0.3]))
        while len(seg) < hom and wi < len(walk):
            x = rng.random()
            if x < div * 0.1:
                wi += 1
                continue
            if x < div * 0.2:
                seg.append(int(rng.choice([65, 67, 71, 84])))
                continue
            b = int(walk[wi])
            wi += 1
            if rng.random() < div:
                b = int(rng.choice([65, 67, 71, 84]))
            seg.append(b)
        while len(seg) < n:
            seg.append(int(rng.choice([65, 67, 71, 84, 78] if case % 11 == 0 else [65, 67, 71, 84])))
        seg = np.array(seg[:n], np.uint8)
        t_ref = C.c_int64(0)
        gbuf = np.ascontiguousarray(genome)
        i_ref = L.orc_ext_align(seg.ctypes.data_as(O.u8p), C.c_int64(n), d, gbuf.ctypes.data_as(O.u8p), C.c_int64(g0), C.c_int64(gmin),
                                C.c_int64(gmax), C.byref(t_ref))
        # the kernel reads the candidate itself: forward (step +1 / -1) or as reverse complement (comp): lay the segment out so
        strand = case % 4 >= 2
        step = +1 if (case // 2) % 2 == 0 else -1
        stored = np.array([comp.get(int(b), 78) for b in seg], np.uint8) if strand else seg.copy()
        q = np.zeros(n + 40, np.uint8)       # (the kernel reads up to three words ahead of the byte it needs)
        if step > 0:
            q[20:20 + n] = stored
            p0 = 20
        else:
            q[20:20 + n] = stored[::-1]
            p0 = 20 + n - 1
        jmax = (gmax - g0) if d > 0 else (g0 - gmin)
        io, to = C.c_int(0), C.c_int(0)
        lib.host_ext(q.ctypes.data_as(O.u8p), C.c_int64(p0), step, int(strand), n, bases.ctypes.data_as(C.POINTER(C.c_uint32)),
                     nm.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(g0), d, C.c_int64(jmax), C.byref(io), C.byref(to))
        assert (io.value, to.value) == (int(i_ref), int(t_ref.value)), (case, n, hom, d, g0, gmin, gmax, io.value, to.value, i_ref, t_ref.value)
        nf = C.c_int(0)
        bad = lib.host_ext_lockstep(q.ctypes.data_as(O.u8p), C.c_int64(p0), step, int(strand), n, bases.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    nm.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(g0), d, C.c_int64(jmax), C.byref(nf))
        assert bad == -1, (case, "bit-parallel and cell-by-cell bands differ at column", bad)
        n_fast_cols += nf.value
        n_cases += 1
        n_cut += 0 < i_ref < n
        n_full += i_ref == n and n > 0
    assert n_cases == 1600 and n_cut > 160 and n_full > 160 and n_fast_cols > 50_000


def test_tr_seed_kernel_logic_vs_twin(tmp_path):
    """the tandem-repeat masker's kernel body (tile load, every thread's scan of its words over the periods, leftmost-of-run rule, extension,
    mask bits) run thread by thread on the host == oracle/hite_oracle_trf.c, on a multi-contig genome with N runs"""
    import casegen
    from test_trmask import twin_mask

    ext = _block(os.path.join(ROOT, "hite_amd", "csrc", "hite_ext.h"), "ext_align_dev")
    body = _block(os.path.join(ROOT, "hite_amd", "csrc", "hite_trmask.hip"), "tr_seed")
    lib = _build(tmp_path, "trseed", "#define EXT_LUT_PTR const uint32_t *\n" + ext + r"""
#define TR_MAXEXT 4096
#define TR_MINSCORE 50
#define TR_RESEED 2048
#define TR_TILE 256
#define TR_HALO 34
static inline void atomicOr(uint32_t *p, uint32_t v) { *p |= v; }
static inline bool tr_defer_push(unsigned long long *, unsigned long long, unsigned long long *, int64_t, int) { return false; }   // (no list on the host)
""" + body, r"""
extern "C" void host_tr_mask(const uint32_t *bases, const uint32_t *nmask, const int64_t *coff, int nc, int64_t G, int max_period, uint32_t *trmask) {
    static TrTile T;
    const int64_t nwords = (G + 15) >> 4;
    for (int64_t w0 = 0; w0 < nwords; w0 += TR_TILE) {
        for (int k = 0; k < TR_TILE + TR_HALO + 2; k++) tr_tile_load(T, k, w0, nwords, G, bases, nmask);
        for (int tid = 0; tid < 256; tid++) tr_thread(T, tid, w0, nwords, G, max_period, bases, nmask, coff, nc, trmask);
    }
}
""")
    seq, _planted = casegen.make_tandem_case(777, G=40_000, n_arr=40)
    contigs = [seq[:9_000], seq[9_000:9_777] + "N" * 40 + seq[9_777:30_003], seq[30_003:], "ACGT" * 10, "ACGTTGCA" * 5]
    genome = np.frombuffer("".join(contigs).encode(), dtype=np.uint8)
    G = len(genome)
    coff = np.zeros(len(contigs) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in contigs], out=coff[1:])
    bases, nm = _pack(genome)
    # the product's buffers carry 8 words of padding; the kernel may read one word beyond the last one it needs
    bases = np.concatenate([bases, np.zeros(40, np.uint32)])
    nm = np.concatenate([nm, np.zeros(40, np.uint32)])
    tr = np.zeros((G + 31) // 32 + 4, dtype=np.uint32)
    lib.host_tr_mask(bases.ctypes.data_as(C.POINTER(C.c_uint32)), nm.ctypes.data_as(C.POINTER(C.c_uint32)), coff.ctypes.data_as(O.i64p),
                     len(contigs), C.c_int64(G), 500, tr.ctypes.data_as(C.POINTER(C.c_uint32)))
    got = np.unpackbits(tr.view(np.uint8), bitorder="little")[:G].astype(bool)
    exp = twin_mask(contigs)
    assert exp.sum() > 3000
    assert np.array_equal(got, exp), (int(got.sum()), int(exp.sum()), np.flatnonzero(got != exp)[:10])


def test_copy_name_order_vs_python_strings(tmp_path):
    """select_rows_kernel's tie-break: copy_name_gt on (contig rank, start, end, strand) == byte order of the window names
    "<contig>:<start>-<end>(<strand>)" (ranks as hite_amd._lib.Context.set_contig_order computes them)"""
    body = _block(os.path.join(ROOT, "hite_amd", "csrc", "hite_pipeline.hip"), "copy_name_cmp")
    lib = _build(tmp_path, "cmp", body, r"""
extern "C" int host_name_gt(int r1, long long s1, long long e1, int m1, int r2, long long s2, long long e2, int m2) {
    return copy_name_gt(r1, s1, e1, m1, r2, s2, e2, m2) ? 1 : 0;
}
""")
    rng = np.random.default_rng(3)
    contigs = ["chr1", "chr10", "chr2", "Chr1", "chr1_random", "1", "10", "X", "scaffold_3"]
    keyed = sorted(range(len(contigs)), key=lambda i: (contigs[i] + ":").encode())
    rank = {contigs[i]: r for r, i in enumerate(keyed)}
    vals = [1, 7, 9, 10, 11, 99, 100, 101, 999, 1000, 1001, 9999, 10000, 99999, 100000, 123456, 1234567, 2_000_000_000, 12345678901]
    def rnd():
        c = contigs[int(rng.integers(0, len(contigs)))]
        s = int(rng.choice(vals)) if rng.random() < 0.7 else int(rng.integers(1, 3_000_000))
        e = int(rng.choice(vals)) if rng.random() < 0.5 else s + int(rng.integers(0, 40000))
        return c, s, e, int(rng.integers(0, 2))
    for _ in range(20000):
        a, b = rnd(), rnd()
        if rng.random() < 0.3:
            b = (a[0], a[1], b[2], b[3])          # same contig and start: decided by the end
        na, nb = ("%s:%d-%d(%s)" % (a[0], a[1], a[2], "+-"[a[3]])).encode(), ("%s:%d-%d(%s)" % (b[0], b[1], b[2], "+-"[b[3]])).encode()
        got = lib.host_name_gt(rank[a[0]], C.c_longlong(a[1]), C.c_longlong(a[2]), a[3], rank[b[0]], C.c_longlong(b[1]), C.c_longlong(b[2]), b[3])
        assert bool(got) == (na > nb), (na, nb, got)


def test_window_decode_vs_slicing(tmp_path):
    """the window gather's per-lane code (hite_genome.h: emit_span with its 16-bases-per-lane body, byte-permute decode, reverse
    complement by bit reversal, head / tail four at a time) == plain slicing of the sequence: every alignment of the destination,
    both strands, spans that start and end anywhere, N bases.  The 64 lanes of the wavefront are run one after the other."""
    body = _block(os.path.join(ROOT, "hite_amd", "csrc", "hite_genome.h"), "genome_decode")
    lib = _build(tmp_path, "decode", r"""
struct uint4 { uint32_t x, y, z, w; };
// v_perm_b32 for selectors 0..3 (bytes of the second operand), which is all the decode uses
static inline uint32_t __builtin_amdgcn_perm(uint32_t a, uint32_t b, uint32_t sel) {
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) { const uint32_t s = (sel >> (8 * i)) & 0xffu; out |= ((s < 4 ? b >> (8 * s) : a >> (8 * (s - 4))) & 0xffu) << (8 * i); }
    return out;
}
static inline uint32_t __builtin_bitreverse32(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
""" + body, r"""
extern "C" void host_emit_span(const uint32_t *bases, const uint32_t *nmask, int64_t g_lo, int64_t wlen, int minus, int64_t ws, int64_t cnt,
                               uint8_t *dst) {
    for (int lane = 0; lane < 64; lane++) emit_span(bases, nmask, g_lo, wlen, minus != 0, ws, cnt, dst, lane);
}
""")
    rng = np.random.default_rng(16)
    comp = np.full(256, ord("N"), np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    n16 = 0
    for case in range(600):
        G = int(rng.integers(300, 6000))
        genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=G)
        if case % 3 == 0:
            genome[rng.integers(0, G, size=int(rng.integers(1, 40)))] = ord("N")
        if case % 10 == 0:
            a = int(rng.integers(0, G - 40)); genome[a:a + 37] = ord("N")
        bases, nm = _pack(genome)
        wlen = int(rng.integers(1, min(G, 3000)))
        g_lo = int(rng.integers(0, G - wlen + 1)) if case % 7 else 0
        minus = case % 2
        ws = int(rng.integers(0, wlen)) if case % 4 else 0
        cnt = int(rng.integers(0, wlen - ws + 1)) if case % 4 else wlen
        window = genome[g_lo:g_lo + wlen]
        if minus:
            window = comp[window[::-1]]
        exp = window[ws:ws + cnt]
        for off in (0, 4, 8, 12, int(rng.integers(0, 16))):          # alignment of the destination within 16 bytes
            buf = np.full(cnt + 64, 0x5a, np.uint8)
            base_addr = buf.ctypes.data
            start = (-base_addr) % 16 + 16 + off
            dst = C.c_void_p(base_addr + start)
            lib.host_emit_span(bases.ctypes.data_as(C.POINTER(C.c_uint32)), nm.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int64(g_lo),
                               C.c_int64(wlen), minus, C.c_int64(ws), C.c_int64(cnt), dst)
            assert np.array_equal(buf[start:start + cnt], exp), (case, off, g_lo, wlen, minus, ws, cnt)
            assert (buf[:start] == 0x5a).all() and (buf[start + cnt:] == 0x5a).all(), (case, off, "wrote outside the span")
            n16 += cnt >= 32
    assert n16 > 1500


def test_fill_sparse_row_vs_twin(tmp_path):
    """hite_fill.h: fill_sparse_row (the body of star_fill_sparse_kernel and of the judge kernels that build their alignment in LDS):
    every row rebuilt from the twin's ops and layout words derived from the twin's FULL alignment + its sparse-column mask must be
    the twin's sparse alignment -- insertion blocks with kept prefixes, dropped centre columns, the extra last column"""
    import casegen
    body = _block(os.path.join(ROOT, "hite_amd", "csrc", "hite_fill.h"), "fill_sparse_row")
    lib = _build(tmp_path, "fill", body, r"""
extern "C" void host_fill(uint8_t *row, const uint8_t *b, int nrow, const uint16_t *rop, const uint32_t *lay, int m, int le, int centre, int ts) {
    if (ts == 1) fill_sparse_row<uint8_t *, 1>(row, b, nrow, rop, lay, m, le, centre != 0, 0);
    else for (int t = 0; t < 64; t++) fill_sparse_row<uint8_t *, 64>(row, b, nrow, rop, lay, m, le, centre != 0, t);
}
""")
    rng = np.random.default_rng(23)
    n_cases = n_ins_kept = n_extra = n_dropped_centre = 0
    for case in range(120):
        m = int(rng.integers(100, 700))
        centre = casegen.rand_seq(rng, m)
        R = int(rng.integers(2, 14))
        rows = [centre]
        shared_ins = casegen.rand_seq(rng, int(rng.integers(1, 6)))
        ins_at = int(rng.integers(5, m - 5))
        for r in range(1, R):
            s = list(casegen.mutate(rng, centre, float(rng.choice([0.0, 0.03, 0.1]))))
            if rng.random() < 0.7:                    # an insertion most rows share: its columns survive sparse-column removal
                s[ins_at:ins_at] = list(shared_ins)
            if rng.random() < 0.75:                   # a deletion most rows share: that centre column is dropped
                del s[40:40 + int(rng.integers(1, 4))]
            if rng.random() < 0.3:                    # a private insertion (sparse: removed) and a tail beyond the centre's end
                k = int(rng.integers(1, len(s) - 1)); s[k:k] = list(casegen.rand_seq(rng, int(rng.integers(1, 4))))
            if case % 3 == 0 and rng.random() < 0.8:
                s += list(casegen.rand_seq(rng, int(rng.integers(1, 5))))
            rows.append("".join(s))
        full, kept = O.star_msa(rows, rows=True)
        if full is None or kept != R:
            continue
        keep = O.sparse_cols(full).astype(bool)
        exp = full[:, keep]
        ops = [None] + [O.align_pair(centre, rows[r])[0] for r in range(1, R)]
        if any(o is None for o in ops[1:]):
            continue

        def ins_of(r, p):
            o = ops[r]
            prev_end = 0 if p == 0 else int(o[p - 1] & 0x7fff) + (0 if (o[p - 1] >> 15) else 1)
            q = int(o[p] & 0x7fff) if p < m else len(rows[r])
            return q - prev_end
        mx = [max(ins_of(r, p) for r in range(1, R)) for p in range(m + 1)]
        fs = np.concatenate([[0], np.cumsum([mx[p] + (1 if p < m else 0) for p in range(m + 1)])])
        assert fs[-1] == full.shape[1]
        before = np.concatenate([[0], np.cumsum(keep)])
        lay = np.zeros(m + 1, np.uint32)
        le = -1
        for p in range(m + 1):
            blk = keep[fs[p]:fs[p] + mx[p]]
            kw = int(blk.sum())
            if p == m and mx[p] > 0 and not blk[:kw].all():
                kw -= 1                               # the forced last column is not part of the prefix
            if p == m and mx[p] > 0 and kw < mx[p] and keep[fs[p] + mx[p] - 1]:
                le = mx[p] - 1
                n_extra += 1
            assert blk[:kw].all() and not blk[kw:mx[p] - (1 if (p == m and le >= 0) else 0)].any(), (case, p)
            keepc = int(keep[fs[p] + mx[p]]) if p < m else 0
            lay[p] = kw | (keepc << 15) | (int(before[fs[p]]) << 16)
            n_ins_kept += kw > 0
            n_dropped_centre += (p < m and not keepc)
        C_out = int(keep.sum())
        for ts in (1, 64):
            for r in range(R):
                out = np.full(C_out + 8, 0x2a, np.uint8)
                b = np.frombuffer(rows[r].encode() + b"\0" * 8, np.uint8)
                # (entry -1 of an ops row = the op "before position 0": a gap op whose q is the row's first base, 0x8000 without pads -- on
                # the device the spare entry of the ops row above, ops_pad_fix_kernel)
                rbuf = np.concatenate([[0x8000], ops[r], [0]]).astype(np.uint16) if r else np.zeros(m + 2, np.uint16)
                rop = C.cast(rbuf.ctypes.data + 2, C.POINTER(C.c_uint16))
                lib.host_fill(out.ctypes.data_as(O.u8p), b.ctypes.data_as(O.u8p), len(rows[r]), rop,
                              lay.ctypes.data_as(C.POINTER(C.c_uint32)), m, le, int(r == 0), ts)
                assert np.array_equal(out[:C_out], exp[r]), (case, r, ts)
                assert (out[C_out:] == 0x2a).all()
        n_cases += 1
    assert n_cases > 80 and n_ins_kept > 50 and n_dropped_centre > 20 and n_extra > 3, (n_cases, n_ins_kept, n_dropped_centre, n_extra)
