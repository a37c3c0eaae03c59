"""CPU: the oracle against golden vectors produced by executing the reference's OWN serialized
TensorFlow graphs (the .meta MetaGraphDefs TF 1.4.1 wrote) node by node with numpy
(tests/meta_graph.py, tests/golden/gen_meta_exec_golden.py).  Wiring, constants, strides,
paddings, concat order and variable names are the reference's; only per-op arithmetic is ours.
Live re-execution of the graphs when /root/reference is present (build container)."""
import os

import numpy as np
import pytest

from conftest import have_reference

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meta_exec_golden.npz")
TOL = 1e-5  # fp32 summation-order noise is ~1e-6; the north star's bar is 1e-4


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


def _luma_from_ctus(ctus):  # [n,64,64] -> one frame n*64 wide
    return np.ascontiguousarray(ctus.transpose(1, 0, 2).reshape(64, -1))


@pytest.mark.parametrize("tag", ["ai_a", "ai_b"])
@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_matches_reference_graph_ai(oracle, gold, tag, mode):
    seed, gain, qp = gold[tag + "_seed_gain_qp"]
    blob = oracle.synth_blob(int(seed), float(gain))
    ctus = gold[tag + "_ctus"]
    F = oracle.features(blob, ctus, mode=mode)
    assert np.abs(F[:8] - gold[tag + "_feat8"]).max() <= TOL
    P, _ = oracle.heads(blob, oracle.fc1(blob, F, mode), int(qp), mode)
    assert np.abs(P - gold[tag + "_probs"]).max() <= TOL
    for thr in (0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8):  # decisions at the shipped thresholds
        far = np.abs(gold[tag + "_probs"] - thr) > TOL
        assert np.array_equal((P > thr)[far], (gold[tag + "_probs"] > thr)[far])


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_matches_reference_graph_ldp_front_end(oracle, gold, mode):
    seed, gain = gold["ldp_seed_gain"]
    blob = oracle.synth_blob(int(seed), float(gain))
    ctus = gold["ldp_ctus"]
    V = oracle.resi_vectors(blob, _luma_from_ctus(ctus), 64 * ctus.shape[0], 64, mode=mode)
    assert np.abs(V - gold["ldp_vec"]).max() <= TOL


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_live_graph_execution_reproduces_the_golden_file(oracle, gold):
    import meta_graph as mg
    nodes = mg.load_nodes(mg.AI_META)
    seed, gain, qp = gold["ai_b_seed_gain_qp"]
    blob = oracle.synth_blob(int(seed), float(gain))
    P, F, ops = mg.run_ai_graph(nodes, dict(oracle.tensor_views(blob)), gold["ai_b_ctus"], int(qp))
    assert np.array_equal(P, gold["ai_b_probs"]) and np.array_equal(F[:8], gold["ai_b_feat8"])
    # the dropout branches (RandomUniform / Floor / RealDiv) are dead at isdrop = 0, as SURVEY 8a says
    assert ops == {"Add", "AvgPool", "ConcatV2", "Conv2D", "Greater", "Identity", "Less", "MatMul", "Maximum", "Mul",
                   "Reshape", "ResizeNearestNeighbor", "Sigmoid", "Sub"}
    # every checkpoint variable of the tensor table is a VariableV2 of the graph, same shape
    for name, view in oracle.tensor_views(blob).items():
        assert nodes[name]["op"] == "VariableV2"
        assert [int(d) for d in nodes[name]["attr"]["shape"]["shape"]] == list(view.shape)
    nodes = mg.load_nodes(mg.LDP_CNN_META)
    seed, gain = gold["ldp_seed_gain"]
    V, _ = mg.run_resi_graph(nodes, dict(oracle.tensor_views(oracle.synth_blob(int(seed), float(gain)))), gold["ldp_ctus"])
    assert np.array_equal(V, gold["ldp_vec"])
