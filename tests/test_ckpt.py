"""CPU: the TF-V2 checkpoint reader (csrc/tf_ckpt_v2.cpp) through the C ABI."""
import os

import numpy as np
import pytest

from conftest import REFERENCE, have_reference
from tfckpt_writer import crc32c, mask, write_bundle

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_index_known_answers(pkg, oracle):
    """The reference's own model_2000000_qp30~35.dat.index (1227-byte data file, copied as a
    fixture) decodes to the tensor table the oracle / kernels are built on."""
    ents = pkg.ethcnn.read_ckpt_index(os.path.join(HERE, "golden", "model_2000000_qp30_35.dat.index"))
    assert len(ents) == 36
    for (name, dtype, shape, shard, off, size, crc), (tname, tshape, toff) in zip(ents, oracle.TENSORS):
        assert (name, dtype, shape, shard, off) == (tname, 1, tuple(tshape), 0, toff)
        assert size == 4 * int(np.prod(tshape))
    assert ents[-1][4] + ents[-1][5] == oracle.BLOB_BYTES == 5152840
    crcs = {e[0]: e[6] for e in ents}
    assert crcs["Variable"] == 3699682245 and crcs["y_conv_flat__64__w"] == 2345433964


def test_lstm_index_known_answers(pkg):
    ents = {e[0]: e for e in pkg.ethcnn.read_ckpt_index(os.path.join(HERE, "golden", "model_LDP_200000_qp32.dat.index"))}
    assert ents["RNN16/fc2/full_connect_b"][2:6] == ((192,), 0, 0, 768)
    assert ents["RNN16/fc2/full_connect_b"][6] == 3777918348
    assert ents["RNN64/multi_rnn_cell/cell_0/lstm_cell/kernel"][2] == (128, 256)
    assert ents["RNN16/multi_rnn_cell/cell_0/lstm_cell/kernel"][2] == (512, 1024)
    assert ents["RNN32/fc3/full_connect_w"][2] == (101, 4)


def test_crc32c(pkg):
    assert crc32c(b"123456789") == 0xE3069283                                      # RFC 3720 check value
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 63, 1000, 4099):
        b = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert pkg.ethcnn.crc32c_masked(b) == mask(crc32c(b))


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_real_lstm_data_crcs(pkg):
    """The one real weight blob in the reference: every tensor's stored crc32c matches."""
    prefix = os.path.join(REFERENCE, "HM-16.5_Test_LDP", "bin", "model_LDP_200000_qp32.dat")
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    ents = pkg.ethcnn.read_ckpt_index(prefix + ".index")
    assert len(ents) == 18 and sum(e[5] for e in ents) == len(data) == 3040312
    for name, dtype, shape, shard, off, size, crc in ents:
        assert pkg.ethcnn.crc32c_masked(data[off:off + size]) == crc, name


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_all_reference_indexes_share_the_layout(pkg, oracle):
    import glob
    paths = glob.glob(os.path.join(REFERENCE, "HM-16.5_Test_AI", "bin", "*.index")) + \
        glob.glob(os.path.join(REFERENCE, "ETH-CNN_Training_AI", "Models", "*.index")) + \
        [os.path.join(REFERENCE, "HM-16.5_Test_LDP", "bin", "model_LDP_2000000_qp22~37.dat.index")]
    seen = 0
    for p in paths:
        if not os.path.exists(p):
            continue
        ents = pkg.ethcnn.read_ckpt_index(p)
        assert [(e[0], e[2], e[4]) for e in ents] == [(n, tuple(s), o) for n, s, o in oracle.TENSORS], p
        seen += 1
    assert seen >= 5


def _tensors(oracle, blob):
    return [(n, np.array(v)) for n, v in oracle.tensor_views(blob).items()]


def test_bundle_round_trip(pkg, oracle, tmp_path):
    blob = oracle.synth_blob(77, 1.0)
    prefix = str(tmp_path / "model_2000000_qp30~35.dat")
    write_bundle(prefix, _tensors(oracle, blob), data_crc=pkg.ethcnn.crc32c_masked)
    got = pkg.ethcnn.read_ckpt_blob(prefix)
    assert np.array_equal(got.view(np.uint32), blob.view(np.uint32))
    ents = pkg.ethcnn.read_ckpt_index(prefix + ".index")
    assert [(e[0], e[4]) for e in ents] == [(n, o) for n, s, o in oracle.TENSORS]


def test_bundle_errors(pkg, oracle, tmp_path):
    E = pkg.EthCnnError
    blob = oracle.synth_blob(78, 1.0)
    p = str(tmp_path / "bad")
    write_bundle(p, _tensors(oracle, blob), data_crc=pkg.ethcnn.crc32c_masked, corrupt="data")
    with pytest.raises(E, match="crc32c mismatch"):
        pkg.ethcnn.read_ckpt_blob(p)
    write_bundle(p, _tensors(oracle, blob), data_crc=pkg.ethcnn.crc32c_masked, corrupt="index_crc")
    with pytest.raises(E, match="crc"):
        pkg.ethcnn.read_ckpt_blob(p)
    write_bundle(p, _tensors(oracle, blob)[:-1], data_crc=pkg.ethcnn.crc32c_masked)
    with pytest.raises(E, match="lacks tensor"):
        pkg.ethcnn.read_ckpt_blob(p)
    t = _tensors(oracle, blob)
    t[3] = (t[3][0], np.zeros(7, np.float32))
    write_bundle(p, t, data_crc=pkg.ethcnn.crc32c_masked)
    with pytest.raises(E, match="shape"):
        pkg.ethcnn.read_ckpt_blob(p)
    with pytest.raises(E):
        pkg.ethcnn.read_ckpt_blob(str(tmp_path / "missing"))
    open(p + ".index", "wb").write(b"\x00" * 100)
    with pytest.raises(E, match="magic"):
        pkg.ethcnn.read_ckpt_index(p + ".index")
    # the reference ships .index without .data: a clear error, not a crash
    import shutil
    shutil.copy(os.path.join(HERE, "golden", "model_2000000_qp30_35.dat.index"), str(tmp_path / "m.index"))
    with pytest.raises(E, match="cannot read"):
        pkg.ethcnn.read_ckpt_blob(str(tmp_path / "m"))
