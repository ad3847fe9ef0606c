"""CPU: the Idefics2 oracle (oracle/idefics2_ref.py) against the fixtures recorded from the reference fork
(tests/golden/make_golden_idefics2.py): NaViT position ids exact, activations / logits / loss / every gradient to fp32 tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle.idefics2_ref import Idefics2Ref, bucketized_position_ids, patch_mask_from_pixel_mask

G = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["idefics2_b1_img2", "idefics2_b1_navit", "idefics2_b2_padimg_rightpad", "idefics2_b1_text_only"]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("case", CASES)
def test_forward_backward_matches_reference(case):
    z = np.load(os.path.join(G, case + ".npz"))
    m = Idefics2Ref.from_npz(os.path.join(G, "weights_idefics2.npz"))
    rec = {}
    pv = z["pixel_values"] if "pixel_values" in z.files else None
    pm = z["pixel_attention_mask"] if "pixel_attention_mask" in z.files else None
    loss, logits = m.forward(z["input_ids"], pv, pm, z["attention_mask"], z["labels"], record=rec)
    loss.backward()
    for k in ("vision_last_hidden_state", "modality_projection_out", "connector_out", "merged_embeds", "llm_layer0_out", "llm_layer1_out"):
        if k in z.files:
            assert np.allclose(rec[k].detach().numpy(), z[k], atol=2e-5, rtol=1e-4), (k, rel_l2(rec[k].detach().numpy(), z[k]))
    am = z["attention_mask"].astype(bool)
    assert np.allclose(logits.detach().numpy()[am], z["logits"][am], atol=2e-5, rtol=1e-4)
    assert abs(float(loss) - float(z["loss"])) <= 1e-6 * abs(float(z["loss"])) + 1e-6
    n = 0
    for k in z.files:
        if k.startswith("grad."):
            g = m.w[k[5:]].grad
            assert g is not None, k
            assert rel_l2(g.numpy(), z[k]) < 1e-4 or np.abs(z[k]).max() < 1e-7, (k, rel_l2(g.numpy(), z[k]))
            n += 1
    assert n >= 21
    assert all(t.grad is None for k, t in m.w.items() if k.startswith("model.vision_model."))


def test_bucketized_position_ids_examples():
    """modeling_idefics2.py:190-210 on hand-checked grids: a full 4x4 image is the identity, a 3x2 image lands on a sub-lattice."""
    full = torch.ones(1, 4, 4, dtype=torch.bool)
    assert bucketized_position_ids(full, 4)[0].tolist() == list(range(16))
    m = torch.zeros(1, 4, 4, dtype=torch.bool)
    m[0, :3, :2] = True
    ids = bucketized_position_ids(m, 4)[0].reshape(4, 4)
    # rows at fractions 0, 1/3, 2/3 -> buckets 0, 1, 2; columns at 0, 1/2 -> buckets 0, 2
    assert ids[:3, :2].tolist() == [[0, 2], [4, 6], [8, 10]]
    assert ids[3].tolist() == [0, 0, 0, 0] and ids[:, 2:].sum() == 0
    pm = torch.zeros(1, 8, 8, dtype=torch.bool)
    pm[0, :5, :3] = True
    assert patch_mask_from_pixel_mask(pm, 2)[0].tolist() == [[True, True, False, False]] * 3 + [[False] * 4]
