"""MCTSNode (/root/reference/self_play.py:19-80) without a GPU: the oracle's tree dump is consistent with its
own search result and with the golden search vectors, and the host mirror's MCTSNode - built from such a dump exactly as it
is built from the device arena - selects, expands and updates like the reference's node."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xq_oracle as xo  # noqa: E402


def _tree(moves, sims):
    oe = xo.OracleEnv()
    for m in moves:
        oe.make_move(xo.encode_move(m))
    mv, vs = oe.search(sims)
    return oe, mv, vs, oe.search_tree(sims)


def test_oracle_tree_dump_is_the_tree_of_the_search():
    for moves, sims in (([], 15), ([], 50), ([(7, 7, 7, 4), (0, 1, 2, 0), (7, 4, 3, 4)], 50), ([(7, 1, 7, 4)], 200)):
        oe, mv, vs, t = _tree(moves, sims)
        n = len(t["move"])
        f, c = int(t["first_child"][0]), int(t["n_child"][0])
        assert t["parent"][0] == -1 and t["visit_count"][0] == sims and c == len(mv) == len(oe.legal_moves())
        assert [int(x) for x in t["move"][f:f + c]] == mv == [int(x) for x in oe.legal_moves()]
        assert [int(x) for x in t["visit_count"][f:f + c]] == vs
        for i in range(n):
            fc, nc = int(t["first_child"][i]), int(t["n_child"][i])
            if nc:
                assert (t["parent"][fc:fc + nc] == i).all()
                # every visit of a node is its own evaluation / terminal value or passes on to exactly one child
                assert t["visit_count"][fc:fc + nc].sum() <= t["visit_count"][i]
                assert abs(float(t["prior"][fc:fc + nc].sum())) > 0
            else:
                assert fc == -1
            if t["visit_count"][i] == 0:
                assert t["value_sum"][i] == 0.0
        # the backup flips the sign at every level: W(parent) = -(sum of the children's W) + the parent's own leaf values
        assert n == 1 + int(t["n_child"].sum())


def test_mcts_node_mirror_on_an_oracle_tree():
    from chinesechessai_amd.self_play import MCTSNode
    from chinesechessai_amd.chess_env import decode_move

    _, mv, vs, t = _tree([(7, 7, 7, 4), (0, 1, 2, 0), (7, 4, 3, 4)], 200)
    arena = {k: t[k] for k in ("visit_count", "value_sum", "prior", "move", "first_child", "n_child")}
    arena["root"] = 0
    root = MCTSNode.from_arena(arena)
    assert root.parent is None and root.move is None and root.visit_count == 200
    assert [decode_move(m) for m in mv] == list(root.children) and vs == [c.visit_count for c in root.children.values()]
    lib = xo.lib()
    seen = [0]

    def walk(node, idx):
        fc, nc = int(t["first_child"][idx]), int(t["n_child"][idx])
        assert len(node.children) == nc and node.is_leaf() == (nc == 0)
        assert node.visit_count == t["visit_count"][idx] and node.value_sum == t["value_sum"][idx]
        assert node.value() == (t["value_sum"][idx] / t["visit_count"][idx] if t["visit_count"][idx] else 0)
        if not nc:
            assert type(node.value_sum) is (int if node.visit_count == 0 else float)     # (the reference's 0 until the first update)
            return
        scores = np.array([lib.xqo_puct_score(float(t["value_sum"][fc + j]), int(t["visit_count"][fc + j]), float(t["prior"][fc + j]),
                                              int(t["visit_count"][idx])) for j in range(nc)], dtype=np.float32)
        m, ch = node.select_child()
        assert ch is list(node.children.values())[int(np.argmax(scores))] and ch.move == m and ch.parent is node
        seen[0] += 1
        for j, c in enumerate(node.children.values()):
            assert isinstance(c.prior_prob, np.float32)
            walk(c, fc + j)

    walk(root, 0)
    assert seen[0] >= 20
    leaf = next(c for c in root.children.values() if c.is_leaf())
    leaf.expand({(0, 0, 1, 0): np.float32(0.5)})
    leaf.expand({(0, 0, 1, 0): np.float32(0.25), (0, 0, 2, 0): np.float32(0.25)})
    assert [c.prior_prob for c in leaf.children.values()] == [np.float32(0.5), np.float32(0.25)]
    n0, w0, rn, rw = leaf.visit_count, leaf.value_sum, root.visit_count, root.value_sum
    leaf.update(-0.25)
    assert (leaf.visit_count, leaf.value_sum, root.visit_count, root.value_sum) == (n0 + 1, w0 - 0.25, rn + 1, rw + 0.25)
    fresh = MCTSNode(prior_prob=np.float32(0.1))
    assert fresh.value() == 0 and fresh.is_leaf() and fresh.select_child() == (None, None)
