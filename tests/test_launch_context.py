"""CPU: `mantis_amd.launch` -- the per-caller launch options that replaced the process-wide switches of round 4 (hip_ops.KERNEL_TIMER,
hip_ops.DW_SUMSQ, mantis_gemm_cu_budget).  Two trainers / models in one process must not see each other's options."""
import threading

import pytest
import torch

from tests import helpers as Hh


def test_context_is_scoped_nested_and_restored_on_exceptions():
    from mantis_amd.launch import LaunchContext, current, launch_context
    base = current()
    assert base.timer is None and base.dw_sumsq is None and base.gemm_cus == 0
    a, b = LaunchContext(timer=[], gemm_cus=200), LaunchContext(gemm_cus=97)
    with launch_context(a):
        assert current() is a
        with launch_context(b):
            assert current() is b
        assert current() is a
        with launch_context(None):                 # None leaves the caller's context in force
            assert current() is a
        with pytest.raises(RuntimeError):
            with launch_context(b):
                raise RuntimeError("step failed")
        assert current() is a
    assert current() is base


def test_context_is_per_thread():
    from mantis_amd.launch import LaunchContext, current, launch_context
    seen = {}

    def worker():
        seen["inside"] = current().gemm_cus
    with launch_context(LaunchContext(gemm_cus=123)):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert current().gemm_cus == 123
    assert seen["inside"] == 0


def test_two_trainers_keep_their_own_options(monkeypatch):
    """Each MantisHipTrainer owns a LaunchContext; the engine installs it for exactly one call.  With the oracle operators in place of the HIP
    backend, a spy on one operator records the context it runs under: trainer A's timer / CU budget never shows up in trainer B's step."""
    import mantis_amd.engine as eng
    from oracle import ops_ref
    from mantis_amd.launch import current
    from mantis_amd.trainer import MantisHipTrainer
    monkeypatch.setattr(eng, "K", ops_ref)
    z = Hh.load_case("siglip_training_step_ga1")
    batch = dict(input_ids=torch.from_numpy(z["mb0.input_ids"]), attention_mask=torch.from_numpy(z["mb0.attention_mask"]),
                 labels=torch.from_numpy(z["mb0.labels"]), pixel_values=Hh.pixels_list(z, "mb0."))
    seen = []
    real = ops_ref.rmsnorm_fwd

    def spy(*a, **k):
        seen.append(current())
        return real(*a, **k)
    monkeypatch.setattr(ops_ref, "rmsnorm_fwd", spy)
    m1, _, _ = Hh.build_product_model("siglip", "cpu")
    m2, _, _ = Hh.build_product_model("siglip", "cpu")

    class Red:                                   # an inactive reducer (world size 1) that carries a CU budget
        active, gemm_cus = False, 200

        def begin(self):
            pass

        def bucket_ready(self, key):
            return ()

        def finish(self):
            pass
    t1, t2 = MantisHipTrainer(m1, 1, reducer=Red()), MantisHipTrainer(m2, 1)
    t1.launch.timer = []
    assert t1.launch.gemm_cus == 200 and t2.launch.gemm_cus == 0
    t1.training_step(m1, batch)
    n1 = len(seen)
    t2.training_step(m2, batch)
    assert n1 > 0 and all(c is t1.launch for c in seen[:n1]) and all(c is t2.launch for c in seen[n1:])
    assert current() is not t1.launch and current() is not t2.launch and current().timer is None
    # a direct engine call (model(**batch), tests) runs without any caller context
    seen.clear()
    m1.engine.step_from_batch(batch, compute_grads=False)
    assert seen and all(c.timer is None and c.gemm_cus == 0 for c in seen)


def test_an_active_reducer_switches_persistent_gemm_workgroups_off():
    """Round 6: a persistent GEMM workgroup that has to wait for a compute unit an RCCL channel holds would walk its whole static tile list late;
    with an ACTIVE gradient reducer the trainer's launches therefore carry `shared_gpu` (flag 32768: one tile per workgroup in the 176-row
    kernel, same results).  No reducer / an inactive one (world size 1): persistent."""
    from mantis_amd.launch import LaunchContext
    from mantis_amd.trainer import MantisHipTrainer
    assert LaunchContext().shared_gpu is False and LaunchContext(shared_gpu=1).shared_gpu is True
    model, _, _ = Hh.build_product_model("siglip", "cpu")

    class Red:
        gemm_cus = 240

        def __init__(self, active):
            self.active = active
    assert MantisHipTrainer(model, 1).launch.shared_gpu is False
    assert MantisHipTrainer(model, 1, reducer=Red(False)).launch.shared_gpu is False
    tr = MantisHipTrainer(model, 1, reducer=Red(True))
    assert tr.launch.shared_gpu is True and tr.launch.gemm_cus == 240
    with pytest.raises(ValueError):
        LaunchContext(gemm_cus=-1)
    with pytest.raises(ValueError):
        LaunchContext(gemm_cus=4096)


def test_side_stream_mode_selection(monkeypatch):
    """hip_ops.side_stream: MANTIS_DW_STREAM = 0 | 1 | low decides; unset, the caller's default does (the fp8 layer loop asks for "low" on a
    single GPU and for the calling stream when bucket hooks are installed, decoder_fp8.decoder_backward).  One stream object per (device, mode)."""
    from mantis_amd import hip_ops as K
    made = []

    class Fake:
        def __init__(self, priority=None):
            self.priority = priority
            made.append(priority)
    monkeypatch.setattr(K, "SideStream", Fake)
    monkeypatch.setattr(K, "_SIDE", {})
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.delenv("MANTIS_DW_STREAM", raising=False)
    assert K.side_stream() is None and K.side_stream(default="0") is None
    low = K.side_stream(default="low")
    assert low is not None and low.priority == "low" and K.side_stream(default="low") is low
    monkeypatch.setenv("MANTIS_DW_STREAM", "0")
    assert K.side_stream(default="low") is None                     # the variable overrides the default
    monkeypatch.setenv("MANTIS_DW_STREAM", "1")
    plain = K.side_stream(default="low")
    assert plain is not None and plain.priority is None and plain is not low
    monkeypatch.setenv("MANTIS_DW_STREAM", "low")
    assert K.side_stream() is low
    assert made == ["low", None]


def test_fp8_backward_asks_for_the_low_priority_stream_only_without_bucket_hooks():
    import inspect
    from mantis_amd import decoder_fp8
    src = inspect.getsource(decoder_fp8.decoder_backward)
    assert 'K.side_stream(default="low" if on_bucket_ready is None else "0")' in src
