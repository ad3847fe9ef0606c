"""CPU: the Qwen2-VL oracle (oracle/qwen2vl_ref.py) against the fixtures recorded from the HF class the reference resolves to
(tests/golden/make_golden_qwen2vl.py): 3-D rope index exact, activations / logits / loss / every gradient to fp32 tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle.qwen2vl_ref import Qwen2VLRef, rope_index, vision_position_ids

G = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["qwen2vl_b1_img2", "qwen2vl_b1_img1_tall", "qwen2vl_b2_rightpad", "qwen2vl_b1_text_only"]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("case", CASES)
def test_forward_backward_matches_reference(case):
    z = np.load(os.path.join(G, case + ".npz"))
    m = Qwen2VLRef.from_npz(os.path.join(G, "weights_qwen2vl.npz"))
    rec = {}
    pv = z["pixel_values"] if "pixel_values" in z.files else None
    grid = z["image_grid_thw"] if "image_grid_thw" in z.files else None
    loss, logits = m.forward(z["input_ids"], pv, grid, z["attention_mask"], z["labels"], record=rec)
    loss.backward()
    if "position_ids" in z.files:
        assert np.array_equal(rec["position_ids"].numpy(), z["position_ids"])          # integer work: bit-exact
    for k in ("vision_patch_embed", "vision_block0_out", "vision_last_hidden_state", "vision_merged", "merged_embeds", "llm_layer0_out",
              "llm_layer1_out"):
        if k in z.files:
            assert np.allclose(rec[k].detach().numpy(), z[k], atol=3e-5, rtol=1e-4), (k, rel_l2(rec[k].detach().numpy(), z[k]))
    am = z["attention_mask"].astype(bool)
    assert np.allclose(logits.detach().numpy()[am], z["logits"][am], atol=3e-5, rtol=1e-4)
    assert abs(float(loss) - float(z["loss"])) <= 1e-6 * abs(float(z["loss"])) + 1e-6
    n = 0
    for k in z.files:
        if k.startswith("grad."):
            g = m.w[k[5:]].grad
            assert g is not None, k
            assert rel_l2(g.numpy(), z[k]) < 1e-4 or np.abs(z[k]).max() < 1e-7, (k, rel_l2(g.numpy(), z[k]))
            n += 1
    assert n == 27                                                                        # 2 layers x 12 + embed, norm, lm_head
    assert all(t.grad is None for k, t in m.w.items() if ".visual." in k)


def test_rope_index_hand_checked():
    """The example of get_rope_index's docstring, images only: text 0..2, a 1x4x4-patch image (2x2 merged tokens), text."""
    IMG = 9
    ids = np.array([[1, 2, 3, IMG, IMG, IMG, IMG, 4, 5]])
    pos = rope_index(ids, None, np.array([[1, 4, 4]]), IMG, 2)
    assert pos[0, 0].tolist() == [0, 1, 2, 3, 3, 3, 3, 5, 6]
    assert pos[1, 0].tolist() == [0, 1, 2, 3, 3, 4, 4, 5, 6]
    assert pos[2, 0].tolist() == [0, 1, 2, 3, 4, 3, 4, 5, 6]
    # left-padded row: masked positions keep 0 and do not advance the counter
    ids = np.array([[0, 0, 1, IMG, IMG, 2]])
    am = np.array([[0, 0, 1, 1, 1, 1]])
    pos = rope_index(ids, am, np.array([[1, 2, 4]]), IMG, 2)
    assert pos[0, 0].tolist() == [0, 0, 0, 1, 1, 3] and pos[2, 0].tolist() == [0, 0, 0, 1, 2, 3]


def test_vision_position_ids_window_order():
    hw = vision_position_ids(np.array([[1, 4, 4]]), 2)
    assert hw[:4].tolist() == [[0, 0], [0, 1], [1, 0], [1, 1]]          # one 2x2 merge group = 4 consecutive rows
    assert hw[4:8].tolist() == [[0, 2], [0, 3], [1, 2], [1, 3]]
    assert vision_position_ids(np.array([[2, 2, 2]]), 2).shape == (8, 2)
