import numpy as np

from analysisgnn_amd.synth import make_batch, make_score_graph, sample_hops


def test_seed0_counts_match_survey():
    g = make_score_graph(seed=0, n_notes=500, add_beats=True, add_measures=True)
    cnt = {et[1]: e.shape[1] for et, e in g.edge_index.items() if et[0] == et[2] == "note"}
    assert cnt == {"onset": 1380, "consecutive": 604, "during": 793, "rest": 528}
    assert g.num_nodes == {"note": 500, "beat": 112, "measure": 28}


def test_edge_rules():
    g = make_score_graph(seed=3, n_notes=120)
    on, du = g.onset_div, g.duration_div
    s, d = g.edge_index[("note", "onset", "note")]
    assert np.all(on[s] == on[d]) and np.sum(s == d) == 120
    s, d = g.edge_index[("note", "consecutive", "note")]
    assert np.all(on[s] + du[s] == on[d])
    s, d = g.edge_index[("note", "during", "note")]
    assert np.all((on[s] < on[d]) & (on[d] < on[s] + du[s]))
    s, d = g.edge_index[("note", "rest", "note")]
    assert np.all(on[s] + du[s] < on[d])


def test_batch_is_block_diagonal():
    b = make_batch(3, n_notes=50, add_beats=True)
    assert b.num_nodes["note"] == 150 and b.num_graphs == 3
    for (s, _, d), e in b.edge_index.items():
        assert np.all(b.batch[s][e[0]] == b.batch[d][e[1]])


def test_sample_hops_is_hop_ordered():
    g = make_score_graph(seed=1, n_notes=200)
    sg = sample_hops(g, n_targets=40, num_neighbors=[5, 5], seed=0)
    npn = sg.num_sampled_nodes["note"]
    assert npn[0] == 40 and sum(npn) == sg.num_nodes["note"]
    bounds = np.cumsum(npn)
    for et, e in sg.edge_index.items():
        per = sg.num_sampled_edges[et]
        assert sum(per) == e.shape[1]
        o = 0
        for hop, c in enumerate(per):          # hop-h edges end in hop-h nodes, start in hops <= h+1
            dst = e[1, o:o + c]
            src = e[0, o:o + c]
            lo = 0 if hop == 0 else bounds[hop - 1]
            assert np.all((dst >= lo) & (dst < bounds[hop]))
            assert np.all(src < bounds[hop + 1])
            o += c
