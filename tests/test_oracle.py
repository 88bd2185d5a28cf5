"""CPU tests: pin the oracle (oracle/bcr_oracle.c) to the reference's golden vectors
(tests/golden/golden.json, produced by the real ropebwt2) and, when oracle/_ref is present,
to the reference itself on fresh random inputs."""
import numpy as np
import pytest

import helpers as H

SO_FLAG = {0: "-LR", 1: "-LRs", 2: "-LRr"}


@pytest.fixture(scope="module", autouse=True)
def _build():
    H.build_oracle()


@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("both", [False, True])
@pytest.mark.parametrize("seq", [False, True])
def test_kat(golden, so, both, seq):
    reads = H.text_to_reads(golden["kat_input"].encode())
    o = H.Oracle(so)
    o.insert_multi(H.encode_batch(reads, True, both), seq=seq)
    flag = SO_FLAG[so] if not both else SO_FLAG[so].replace("R", "")
    assert H.bwt_text(o.bwt()).decode() == golden["kat"][flag]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_kat_insert1(golden, so):
    """mr_insert1 restatement (-m0 path) gives the same BWT as the batch path."""
    reads = H.text_to_reads(golden["kat_input"].encode())
    o = H.Oracle(so)
    for r in reads:
        o.insert1(r[::-1])
    assert H.bwt_text(o.bwt()).decode() == golden["kat"][SO_FLAG[so]]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_golden_10k_text_md5(golden, so):
    g = golden["sets"]["10k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    assert H.md5(H.reads_to_text(codes)) == g["input_md5"]
    o = H.Oracle(so)
    o.insert_multi(H.encode_batch_fixed(codes))
    assert H.md5(H.bwt_text(o.bwt()) + b"\n") == g["text_md5"][SO_FLAG[so]]


def test_golden_10k_both_strands(golden):
    g = golden["sets"]["10k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    o = H.Oracle(2)
    o.insert_multi(H.encode_batch_fixed(codes, True, True))
    assert H.md5(H.bwt_text(o.bwt()) + b"\n") == g["text_md5"]["-Lr"]


@pytest.mark.parametrize("so", [0, 1, 2])
@pytest.mark.parametrize("seed", [1, 2])
def test_golden_repetitive(golden, so, seed):
    g = golden["sets"]["rep1500_seed%d" % seed]
    reads = H.repetitive_reads(1500, seed=seed)
    assert H.md5(H.lines_from_codes(reads)) == g["input_md5"]
    for nb in (1, 3):
        o = H.Oracle(so)
        per = (len(reads) + nb - 1) // nb
        for i in range(0, len(reads), per):
            o.insert_multi(H.encode_batch(reads[i:i + per]))
        assert H.md5(H.bwt_text(o.bwt()) + b"\n") == g["text"][SO_FLAG[so]]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_seq_equals_bulk(so):
    reads = H.repetitive_reads(400, seed=20 + so, genome_len=90, max_len=25)
    a, b = H.Oracle(so), H.Oracle(so)
    for i in range(0, 400, 150):
        buf = H.encode_batch(reads[i:i + 150], True, so == 2)
        a.insert_multi(buf, seq=True)
        b.insert_multi(buf, seq=False)
    assert np.array_equal(a.bwt(), b.bwt())
    assert np.array_equal(a.counts(), b.counts())


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("so", [0, 1, 2])
def test_against_reference_binary(so):
    rng = np.random.RandomState(100 + so)
    reads = [rng.randint(1, 6, size=rng.randint(0, 60)).astype(np.uint8) for _ in range(700)]
    text = H.lines_from_codes(reads)
    ref = H.run_ref([SO_FLAG[so]], text).strip()
    o = H.Oracle(so)
    o.insert_multi(H.encode_batch(reads[:300]))
    o.insert_multi(H.encode_batch(reads[300:]))
    assert H.bwt_text(o.bwt()) == ref


def test_incremental_equals_oneshot():
    """one-shot build == build of first half then insertion of second half."""
    codes = H.splitmix_bases(2000, 50, seed=9)
    for so in (0, 1, 2):
        a, b = H.Oracle(so), H.Oracle(so)
        a.insert_multi(H.encode_batch_fixed(codes))
        b.insert_multi(H.encode_batch_fixed(codes[:1000]))
        b.insert_multi(H.encode_batch_fixed(codes[1000:]))
        assert np.array_equal(a.bwt(), b.bwt())


def test_codec_roundtrip():
    import ctypes as C
    L = H.oracle_lib()
    buf = (C.c_uint8 * 16)()
    c, l = C.c_int(), C.c_int64()
    for length in [1, 2, 15, 16, 17, 255, 256, 1000, (1 << 19) - 1, 1 << 19, (1 << 30) + 12345, (1 << 43) - 1]:
        for sym in range(6):
            n = L.orc_rle_enc1(buf, sym, length)
            assert n == (1 if length < 16 else 2 if length < 256 else 4 if length < (1 << 19) else 8)
            m = L.orc_rle_dec1(buf, C.byref(c), C.byref(l))
            assert (m, c.value, l.value) == (n, sym, length)


def test_kat_fmr_leaf_bytes(golden):
    """The tiny .fmr of the reference carries these run bytes per rope; our codec
    restatement must produce the same bytes for the same runs."""
    import ctypes as C
    L = H.oracle_lib()
    expect = {0: "0b090b09080a", 1: "0c0d10", 2: "0811", 3: "1208", 4: "0c08", 5: "0b"}
    reads = H.text_to_reads(golden["kat_input"].encode())
    o = H.Oracle(0)
    o.insert_multi(H.encode_batch(reads))
    for b, hx in expect.items():
        r = o.rope(b)
        out = bytearray()
        i = 0
        buf = (C.c_uint8 * 8)()
        while i < len(r):
            j = i
            while j < len(r) and r[j] == r[i]:
                j += 1
            n = L.orc_rle_enc1(buf, int(r[i]), j - i)
            out += bytes(buf[:n])
            i = j
        assert out.hex() == hx
        assert hx in golden["kat_fmr_hex"]
