""".bed ingest through the C ABI (gmrm_load_bed_file) -- validation the reference lacks
(magic bytes, file size) and byte-exact placement of a marker block, incl. ragged N and a block that
starts in the middle of the file (a rank's shard, bayes.cpp:867-900)."""
import numpy as np
import pytest

import gmrm_amd
from gmrm_amd import io

pytestmark = pytest.mark.gpu


def _cols(rng, M, N):
    mb = (N + 3) // 4
    cols = rng.integers(0, 256, size=(M, mb), dtype=np.uint8)
    if N % 4:
        cols[:, -1] &= (1 << (2 * (N % 4))) - 1          # pad bits 0, as PLINK writes them
    return cols


@pytest.mark.parametrize("N,M,first,block,threads", [(1003, 300, 0, 300, 1), (4096, 2000, 700, 900, 8), (50_001, 3000, 0, 3000, 16)])
def test_load_bed_file_places_the_block_byte_for_byte(tmp_path, N, M, first, block, threads):
    rng = np.random.default_rng(5)
    cols = _cols(rng, M, N)
    path = tmp_path / "x.bed"
    io.write_bed(path, cols)
    ctx = gmrm_amd.Context(N, block, Mt=M, S=first, T=1)
    try:
        st = ctx.load_bed_file(path, threads=threads)
        assert st["bytes"] == block * ((N + 3) // 4)
        got = ctx.download_bed()
        assert np.array_equal(got, cols[first:first + block])
    finally:
        ctx.close()


def test_load_bed_file_rejects_bad_magic_and_short_files(tmp_path):
    N, M = 1000, 64
    rng = np.random.default_rng(6)
    cols = _cols(rng, M, N)
    good = tmp_path / "good.bed"
    io.write_bed(good, cols)
    raw = good.read_bytes()
    bad = tmp_path / "bad.bed"
    bad.write_bytes(b"\x6c\x1b\x00" + raw[3:])           # individual-major flag: the reference would read it as SNP-major
    short = tmp_path / "short.bed"
    short.write_bytes(raw[:-17])
    ctx = gmrm_amd.Context(N, M, T=1)
    try:
        with pytest.raises(gmrm_amd.GmrmError, match="magic"):
            ctx.load_bed_file(bad)
        with pytest.raises(gmrm_amd.GmrmError, match="shorter"):
            ctx.load_bed_file(short)
        with pytest.raises(gmrm_amd.GmrmError, match="cannot open"):
            ctx.load_bed_file(tmp_path / "missing.bed")
        ctx.load_bed_file(good)
        assert np.array_equal(ctx.download_bed(), cols)
    finally:
        ctx.close()
