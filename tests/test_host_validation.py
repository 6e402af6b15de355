"""CPU: the product's host layer (JSON reader, model / reduction validation) accepts
and rejects the same inputs as the reference, and as the oracle's restatement.

Cases restate the reference's test_scripts/test_bad_model_args.py,
test_bad_reduction_args.py and test_ok_model_args.py (same inputs, same
expectation: the reference raises RuntimeError for every "bad" case).  No GPU
is touched: arbplf_validate_string() stops after validation."""
import copy
import ctypes
import json

import pytest

GOOD = {
    "model_and_data": {
        "edges": [[0, 1], [1, 2], [1, 3]],
        "edge_rate_coefficients": [2.0, 4.2, 0.5],
        "rate_matrix": [[0, 4.2, 3.0], [1.0, 0, 5.0], [6.0, 0.5, 0]],
        "probability_array": [
            [[0.6, 0.2, 0.2], [1, 1, 1], [1, 0, 0], [0, 0, 1]],
            [[0.6, 0.2, 0.2], [1, 1, 1], [0, 0, 1], [0, 0, 1]]]},
    "site_reduction": {"aggregation": "sum"},
}


@pytest.fixture(scope="module")
def validate():
    from phyly_amd import engine
    lib = engine.load_library()
    lib.arbplf_validate_string.argtypes = [ctypes.c_char_p, ctypes.c_char_p]

    def f(obj, kind="ll"):
        s = obj if isinstance(obj, str) else json.dumps(obj)
        return lib.arbplf_validate_string(kind.encode(), s.encode())
    return f


def _oracle_accepts(oracle, obj, kind="ll"):
    fn = {"ll": oracle.run_ll, "deriv": oracle.run_deriv, "marginal": oracle.run_marginal,
          "dwell": oracle.run_dwell, "trans": oracle.run_trans, "em_update": oracle.run_em_update,
          "hess": oracle.run_hess}[kind]
    try:
        fn(copy.deepcopy(obj))
        return True
    except oracle.OracleError:
        return False


def _md(**kw):
    x = copy.deepcopy(GOOD)
    x["model_and_data"].update(kw)
    return x


def _pa(i, j=None, k=None):
    x = copy.deepcopy(GOOD)
    p = x["model_and_data"]["probability_array"]
    if j is None:
        p[i] = "foo"
    elif k is None:
        p[i][j] = "foo"
    else:
        p[i][j][k] = "foo"
    return x


def _edge(idx, val):
    x = copy.deepcopy(GOOD)
    x["model_and_data"]["edges"][idx] = val
    return x


BAD_MODELS = {
    "edge_dict": _edge(2, {"hello": "world"}),
    "edge_string": _edge(2, "hello"),
    "edge_int": _edge(2, 0),
    "edge_negative_node": _edge(2, [-1, 3]),
    "edge_too_high_node": _md(edges=[[0, 1], [1, 5], [1, 6]]),
    "edge_long": _edge(0, [0, 1, 2]),
    "edge_short": _edge(0, [0]),
    "edge_float_index": _edge(0, [0.0, 1]),
    "edge_connected_loop": _md(edges=[[0, 1], [1, 2], [2, 2]]),
    "edge_disconnected_loop": _md(edges=[[0, 1], [1, 2], [3, 3]]),
    "edges_dag_not_tree": _md(edges=[[0, 1], [1, 2], [1, 3], [2, 3]]),
    "edges_cycle": _md(edges=[[0, 1], [1, 2], [2, 0]]),
    "edges_disconnected_cycle": _md(edges=[[0, 1], [1, 2], [2, 0], [3, 4], [4, 5], [4, 6]]),
    "edges_undirected_tree": _md(edges=[[0, 1], [2, 1], [1, 3]]),
    "edges_disconnected_dag_and_tree": _md(edges=[[0, 1], [0, 2], [1, 3], [2, 3], [4, 5]]),
    "coeffs_not_array": _md(edge_rate_coefficients={"hello": "world"}),
    "coeffs_too_many": _md(edge_rate_coefficients=[1, 2, 3, 4]),
    "coeffs_too_few": _md(edge_rate_coefficients=[1, 2]),
    "coeffs_string": _md(edge_rate_coefficients=[1, 2, "wat"]),
    "coeffs_number_as_string": _md(edge_rate_coefficients=[1, 2, "3"]),
    "coeffs_negative": _md(edge_rate_coefficients=[-1, 2, 3]),
    "rate_matrix_not_array": _md(rate_matrix={"hello": "world"}),
    "rate_matrix_row_not_array": _md(rate_matrix=[[0, 4.2, 3.0], {"hello": "world"}, [6.0, 0.5, 0]]),
    "rate_matrix_row_too_long": _md(rate_matrix=[[0, 4.2, 3.0], [1.0, 0, 5.0, 8.0], [6.0, 0.5, 0]]),
    "rate_matrix_row_too_short": _md(rate_matrix=[[0, 4.2, 3.0], [1.0, 0], [6.0, 0.5, 0]]),
    "rate_matrix_string_entry": _md(rate_matrix=[[0, 4.2, 3.0], [1.0, "wat", 5.0], [6.0, 0.5, 0]]),
    "rate_matrix_number_as_string": _md(rate_matrix=[[0, 4.2, 3.0], [1.0, "42", 5.0], [6.0, 0.5, 0]]),
    "rate_matrix_negative": _md(rate_matrix=[[0, 4.2, 3.0], [1.0, 0, -5.0], [6.0, 0.5, 0]]),
    "prob_level0": _md(probability_array="foo"),
    "prob_level1a": _pa(0), "prob_level1b": _pa(1),
    "prob_level2a": _pa(0, 0), "prob_level2b": _pa(1, 1),
    "prob_level3a": _pa(0, 0, 0), "prob_level3b": _pa(1, 1, 1),
    "prob_too_many_nodes": _md(probability_array=[
        [[0.6, 0.2, 0.2], [1, 1, 1], [1, 0, 0], [0, 0, 1], [1, 1, 1]],
        [[0.6, 0.2, 0.2], [1, 1, 1], [0, 0, 1], [0, 0, 1], [1, 1, 1]]]),
    "prob_too_few_nodes": _md(probability_array=[
        [[0.6, 0.2, 0.2], [0, 0, 1], [1, 1, 1]], [[0.6, 0.2, 0.2], [0, 0, 1], [1, 1, 1]]]),
    "prob_too_many_states": _md(probability_array=[
        [[0.6, 0.2, 0.2, 0.1], [1, 1, 1, 1], [1, 0, 0, 1], [0, 0, 1, 1]],
        [[0.6, 0.2, 0.2, 0.1], [1, 1, 1, 1], [0, 0, 1, 1], [0, 0, 1, 1]]]),
    "prob_too_few_states": _md(probability_array=[
        [[0.6, 0.2], [1, 1], [1, 0], [0, 0]], [[0.6, 0.2], [1, 1], [0, 0], [0, 0]]]),
    # further grammar the parser enforces (src/parsemodel.c)
    "unknown_model_key": _md(hello="world"),
    "unknown_top_key": dict(copy.deepcopy(GOOD), hello=1),
    "missing_data": {"model_and_data": {k: v for k, v in GOOD["model_and_data"].items() if k != "probability_array"}},
    "prob_negative": _md(probability_array=[[[0.6, -0.2, 0.2], [1, 1, 1], [1, 0, 0], [0, 0, 1]]]),
    "both_data_forms": _md(character_data=[[0, 0, 0, 0]], character_definitions=[[1, 1, 1]]),
    "rate_divisor_zero": _md(rate_divisor=0),
    "rate_divisor_bad_string": _md(rate_divisor="exit_rate"),
    "root_prior_bad_string": _md(root_prior="stationary"),
    "root_prior_wrong_length": _md(root_prior=[0.5, 0.5]),
    "two_mixtures": _md(rate_mixture={"rates": [1, 2], "prior": [0.5, 0.5]},
                        gamma_rate_mixture={"gamma_shape": 1.0, "gamma_categories": 4}),
    "mixture_prior_length": _md(rate_mixture={"rates": [1, 2], "prior": [0.5, 0.25, 0.25]}),
    "mixture_prior_bad_string": _md(rate_mixture={"rates": [1, 2], "prior": "uniform"}),
    "gamma_categories_float": _md(gamma_rate_mixture={"gamma_shape": 1.0, "gamma_categories": 4.0}),
    "gamma_unknown_key": _md(gamma_rate_mixture={"gamma_shape": 1.0, "gamma_categories": 4, "x": 1}),
    "bool_is_not_a_number": _md(edge_rate_coefficients=[True, 2, 3]),
}


def _chardata(**kw):
    x = copy.deepcopy(GOOD)
    md = x["model_and_data"]
    del md["probability_array"]
    md["character_definitions"] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]]
    md["character_data"] = [[0, 3, 1, 2], [2, 3, 2, 2]]
    md.update(kw)
    return x


BAD_MODELS.update({
    "chardata_without_definitions": {"model_and_data": {k: v for k, v in _chardata()["model_and_data"].items()
                                                         if k != "character_definitions"}},
    "chardata_float_code": _chardata(character_data=[[0, 3.0, 1, 2]]),
    "chardata_code_out_of_range": _chardata(character_data=[[0, 4, 1, 2]]),
    "chardata_negative_code": _chardata(character_data=[[0, -1, 1, 2]]),
    "chardata_wrong_node_count": _chardata(character_data=[[0, 1, 2]]),
    "chardefs_wrong_state_count": _chardata(character_definitions=[[1, 0], [0, 1], [1, 1], [1, 1]]),
})


@pytest.mark.parametrize("name", sorted(BAD_MODELS))
def test_bad_model_rejected(validate, oracle, name):
    x = BAD_MODELS[name]
    assert validate(x) != 0
    assert not _oracle_accepts(oracle, x)


def _red(**kw):
    x = copy.deepcopy(GOOD)
    x["site_reduction"].update(kw)
    return x


BAD_REDUCTIONS = {
    "reduction_unknown_key": _red(hello="world"),
    "aggregation_dict": _red(aggregation={"hello": "world"}),
    "aggregation_bad_string": _red(aggregation="foo"),
    "aggregation_weight_strings": _red(aggregation=["42", "3"]),
    "aggregation_too_few_weights": _red(aggregation=[42.0]),
    "aggregation_too_many_weights": _red(aggregation=[42.0, 43.0, 44.0]),
    "selection_dict": _red(selection={"hello": "world"}),
    "selection_string": _red(selection="hello"),
    "selection_negative": _red(selection=[0, -2]),
    "selection_too_large": _red(selection=[100, 0]),
    "selection_float": _red(selection=[3.14]),
    "selection_string_index": _red(selection=["0"]),
    "selection_dict_index": _red(selection=[{"hello": "world"}]),
    "selection_too_few_weights_2": _red(selection=[0, 1, 0, 1], aggregation=[0.1, 0.2]),
    "selection_too_few_weights_3": _red(selection=[0, 1, 0, 1], aggregation=[0.1, 0.2, 0.3]),
    "selection_too_many_weights": _red(selection=[0, 1, 0, 1], aggregation=[0.1, 0.2, 0.3, 0.4, 0.5]),
    "only_needs_one_selected": _red(selection=[0, 1], aggregation="only"),
    "reduction_null": dict(copy.deepcopy(GOOD), site_reduction=None),
    "selection_null": _red(selection=None),
}


@pytest.mark.parametrize("name", sorted(BAD_REDUCTIONS))
def test_bad_reduction_rejected(validate, oracle, name):
    x = BAD_REDUCTIONS[name]
    assert validate(x) != 0
    assert not _oracle_accepts(oracle, x)


def _ok_cases():
    yield "good", GOOD
    x = copy.deepcopy(GOOD); del x["site_reduction"]; yield "reduction_deleted", x
    yield "reduction_empty", dict(copy.deepcopy(GOOD), site_reduction={})
    yield "avg", dict(copy.deepcopy(GOOD), site_reduction={"aggregation": "avg"})
    yield "selection", dict(copy.deepcopy(GOOD), site_reduction={"selection": [0]})
    yield "selection_sum", dict(copy.deepcopy(GOOD), site_reduction={"selection": [0], "aggregation": "sum"})
    yield "selection_only", dict(copy.deepcopy(GOOD), site_reduction={"selection": [1], "aggregation": "only"})
    yield "duplicates_weighted", dict(copy.deepcopy(GOOD), site_reduction={"selection": [1, 1, 0], "aggregation": [1, -1, 2.5]})
    yield "chardata", _chardata()
    yield "null_optional_keys", _md(rate_divisor=None, root_prior=None, rate_mixture=None)
    yield "divisor_string", _md(rate_divisor="equilibrium_exit_rate", root_prior="equilibrium_distribution")
    yield "uniform_root", _md(root_prior="uniform_distribution")
    yield "custom_root_int_entries", _md(root_prior=[1, 0, 0])
    yield "mixture_uniform", _md(rate_mixture={"rates": [0, 1, 2.5], "prior": "uniform_distribution"})
    yield "gamma_inv", _md(gamma_rate_mixture={"gamma_shape": 0.5, "gamma_categories": 4, "invariable_prior": 0.3})
    yield "median_gamma", _md(normalized_median_gamma_rate_mixture={"gamma_shape": 2, "gamma_categories": 3})
    yield "simplified_fig_16_4", {
        "model_and_data": {
            "edges": [[5, 0], [5, 1], [5, 6], [6, 2], [6, 7], [7, 3], [7, 4]],
            "edge_rate_coefficients": [0.01, 0.2, 0.15, 0.3, 0.05, 0.3, 0.02],
            "rate_matrix": [[0, 3, 3, 3], [3, 0, 3, 3], [3, 3, 0, 3], [3, 3, 3, 0]],
            "probability_array": [[[1, 0, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0],
                                   [0.25, 0.25, 0.25, 0.25], [1, 1, 1, 1], [1, 1, 1, 1]]]},
        "site_reduction": {"aggregation": "sum"}}


OK = dict(_ok_cases())


@pytest.mark.parametrize("name", sorted(OK))
def test_ok_model_accepted(validate, oracle, name):
    assert validate(OK[name]) == 0
    assert _oracle_accepts(oracle, OK[name])


def test_deriv_and_marginal_reductions(validate, oracle):
    x = copy.deepcopy(GOOD)
    x["edge_reduction"] = {"selection": [2, 0], "aggregation": [1.5, -1]}
    assert validate(x, "deriv") == 0 and _oracle_accepts(oracle, x, "deriv")
    assert validate(x, "ll") != 0                      # edge_reduction is unknown to arbplf-ll
    assert validate(x, "marginal") != 0
    y = copy.deepcopy(GOOD)
    y["node_reduction"] = {"selection": [3]}
    y["state_reduction"] = {"aggregation": "sum"}
    assert validate(y, "marginal") == 0 and _oracle_accepts(oracle, y, "marginal")
    assert validate(y, "deriv") != 0
    y["state_reduction"] = {"selection": [3]}         # only 3 states
    assert validate(y, "marginal") != 0 and not _oracle_accepts(oracle, y, "marginal")
    x["edge_reduction"] = {"selection": [3]}           # only 3 edges
    assert validate(x, "deriv") != 0 and not _oracle_accepts(oracle, x, "deriv")


def test_dwell_trans_em_reductions(validate, oracle):
    """src/arbplfdwell.c:507-566, src/arbplftrans.c:553-614 (pair selection: src/parsereduction.c:205-392),
    src/arbplfem.c:505-545 + :572-577 (site aggregation required)"""
    def both(x, kind, ok):
        assert (validate(x, kind) == 0) == ok, (kind, x)
        assert _oracle_accepts(oracle, x, kind) == ok, (kind, x)
    x = copy.deepcopy(GOOD)
    x.pop("site_reduction", None)
    both(x, "dwell", True)
    both(x, "trans", True)
    both(x, "em_update", False)                         # needs a site aggregation
    both(x, "hess", False)                              # site_reduction itself is required (src/arbplfhess.c:1178-1182)
    x["site_reduction"] = {"selection": [0]}
    both(x, "hess", False)                              # ... and must aggregate
    x["site_reduction"] = {"aggregation": "sum"}
    both(x, "em_update", True)
    both(x, "hess", True)
    x["edge_reduction"] = {"aggregation": "sum"}
    both(x, "em_update", False)                         # edge_reduction is unknown to em-update
    both(x, "dwell", True)
    x["state_reduction"] = {"selection": [0, 2, 2], "aggregation": [1, 2, -0.5]}
    both(x, "dwell", True)
    both(x, "trans", False)                             # state_reduction is unknown to arbplf-trans
    del x["state_reduction"]
    for tr, ok in (({"selection": [[0, 1], [2, 0]]}, True),
                   ({"selection": [[0, 1], [2, 0]], "aggregation": [0.5, 2]}, True),
                   ({"selection": [[0, 1]], "aggregation": "only"}, True),
                   ({"selection": [[0, 1], [1, 0]], "aggregation": "only"}, False),
                   ({"selection": [[1, 1]]}, True),                      # a diagonal pair is accepted
                   ({"selection": [[0, 3]]}, False),                     # only 3 states
                   ({"selection": [[0, -1]]}, False),
                   ({"selection": [[0, 1, 2]]}, False),
                   ({"selection": [[0, 1.0]]}, False),
                   ({"selection": [0, 1]}, False),
                   ({"selection": [[0, 1]], "aggregation": [1, 2]}, False),
                   ({"aggregation": "sum"}, True),
                   ({"aggregation": "avg"}, True),
                   ({"aggregation": "only"}, False),                     # no selection: only sum / avg
                   ({"aggregation": [1, 2, 3, 4, 5, 6]}, False),
                   ({}, True),
                   ({"selection": None}, True),                          # null counts as absent
                   ({"selection": [], "aggregation": "sum"}, True),
                   ({"selektion": []}, False)):
        y = copy.deepcopy(x)
        y["trans_reduction"] = tr
        both(y, "trans", ok)
        assert validate(y, "dwell") != 0


@pytest.mark.parametrize("text", [
    "", "   ", "42", '"string"', "{", "[1, 2", '{"a": 1,}', "{'a': 1}", '{"a": 01}', '{"a": 1.}', '{"a": .5}',
    '{"a": +1}', '{"a": NaN}', '{"a": Infinity}', '{"a": 1e999}', '{"a": "\\x"}', '{"a": "tab\there"}',
    '{"a": 1} trailing', '{"a": 99999999999999999999}', '{"a" 1}', '[1 2]', '{"a": tru}'])
def test_malformed_json_rejected(validate, text):
    assert validate(text) != 0


def test_json_numbers_and_unicode(validate):
    x = copy.deepcopy(GOOD)
    s = json.dumps(x).replace("4.2", "4.2e0").replace('"sum"', '"\\u0073um"')
    assert validate(s) == 0
    s2 = json.dumps(x).replace('"edges"', '"edges": [[0, 1]], "edges"')   # duplicate key: last wins
    assert validate(s2) == 0


def test_character_data_file_side_channel(validate, tmp_path):
    """binary side channel for large alignments (an extension of the reference's input):
    'character_data_file' = raw [site][node] bytes, used with 'character_definitions'"""
    x = copy.deepcopy(GOOD)
    md = x["model_and_data"]
    md.pop("probability_array", None)
    md.pop("character_data", None)
    N = len(md["edges"]) + 1
    k = len(md["rate_matrix"])
    md["character_definitions"] = [[1.0 if j == c else 0.0 for j in range(k)] for c in range(k)] + [[1.0] * k]
    codes = bytes([(s * 7 + a) % (k + 1) for s in range(5) for a in range(N)])
    f = tmp_path / "codes.u8"
    f.write_bytes(codes)
    md["character_data_file"] = str(f)
    x.pop("site_reduction", None)
    assert validate(x) == 0
    bad = copy.deepcopy(x)
    bad["model_and_data"]["character_data_file"] = str(tmp_path / "missing.u8")
    assert validate(bad) != 0
    (tmp_path / "ragged.u8").write_bytes(codes[:-1])
    bad["model_and_data"]["character_data_file"] = str(tmp_path / "ragged.u8")
    assert validate(bad) != 0
    (tmp_path / "range.u8").write_bytes(codes[:-1] + bytes([k + 1]))
    bad["model_and_data"]["character_data_file"] = str(tmp_path / "range.u8")
    assert validate(bad) != 0
    bad = copy.deepcopy(x)
    bad["model_and_data"]["character_data"] = [[0] * N]
    assert validate(bad) != 0                       # both forms at once
    bad = copy.deepcopy(x)
    del bad["model_and_data"]["character_definitions"]
    assert validate(bad) != 0
    bad = copy.deepcopy(x)
    bad["model_and_data"]["character_data_file"] = 17
    assert validate(bad) != 0


def test_empty_edge_list_is_rejected_like_the_reference(validate, capfd, oracle):
    """`edges: []` gives node_count = 1 in the reference (src/parsemodel.c:230-231) and is then REJECTED by its own
    degree check, "node index 0 is not an endpoint of any edge" (src/parsemodel.c:306-318): the reference has no
    single-node tree.  Product and oracle reject it the same way (exit status / RuntimeError)."""
    x = {"model_and_data": {"edges": [], "edge_rate_coefficients": [],
                            "rate_matrix": [[0, 1.0], [1.0, 0]],
                            "probability_array": [[[1, 0]]]},
         "site_reduction": {"aggregation": "sum"}}
    assert validate(x) != 0
    assert "not an endpoint of any edge" in capfd.readouterr().err
    import json
    with pytest.raises(RuntimeError):
        oracle.arbplf_ll(json.dumps(x))
