"""Host side of the serve / data path against the reference:
  1. the reference's own known-answer tests, restated against this package (transforms_test.py:8-121,
     image_tools_test.py:6-37, msgpack_numpy_test.py:17-45);
  2. tests/golden/host_pipeline.npz — outputs of the reference's code EXECUTED on seeded inputs
     (tests/golden/make_host_pipeline_golden.py): image resize, wire bytes, every numeric transform, the pi0 / pi0.5 prompt
     tokenisation, the Agilex robot transforms and the whole `Policy.infer` flow.  Integer / byte results must be identical,
     float64 transforms equal to the last bit (same numpy expressions)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kai0_amd import agilex_policy, image_tools, msgpack_numpy, policy, tokenizer, transforms  # noqa: E402
from kai0_amd.normalize import NormStats  # noqa: E402

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_pipeline.npz"))


# ------------------------------------------------------------------------------------- 1. the reference's known-answer tests
def test_repack_transform():
    t = transforms.RepackTransform(structure={"a": {"b": "b/c"}, "d": "e/f"})
    assert t({"b": {"c": 1}, "e": {"f": 2}}) == {"a": {"b": 1}, "d": 2}


def test_delta_and_absolute_actions():
    item = {"state": np.array([1, 2, 3]), "actions": np.array([[3, 4, 5], [5, 6, 7]])}
    out = transforms.DeltaActions(mask=[False, True])(item)
    assert np.all(out["state"] == np.array([1, 2, 3])) and np.all(out["actions"] == np.array([[3, 2, 5], [5, 4, 7]]))
    item = {"state": np.array([1, 2, 3]), "actions": np.array([[3, 4, 5], [5, 6, 7]])}
    out = transforms.AbsoluteActions(mask=[False, True])(item)
    assert np.all(out["state"] == np.array([1, 2, 3])) and np.all(out["actions"] == np.array([[3, 6, 5], [5, 8, 7]]))


@pytest.mark.parametrize("cls", [transforms.DeltaActions, transforms.AbsoluteActions])
def test_action_transforms_noop(cls):
    item = {"state": np.array([1, 2, 3]), "actions": np.array([[3, 4, 5], [5, 6, 7]])}
    assert cls(mask=None)(item) is item
    del item["actions"]
    assert cls(mask=[True, False])(item) is item


def test_make_bool_mask():
    assert transforms.make_bool_mask(2, -2, 2) == (True, True, False, False, True, True)
    assert transforms.make_bool_mask(2, 0, 2) == (True, True, True, True)


def test_transform_dict():
    inp = {"a": {"b": 1, "c": 2}}
    assert transforms.transform_dict({"a/b": "a/c", "a/c": None}, inp) == {"a": {"c": 1}}
    with pytest.raises(ValueError, match="Key 'a/c' already exists in output"):
        transforms.transform_dict({"a/b": "a/c"}, inp)
    assert transforms.transform_dict({"a": None}, inp) == inp  # a full match is required
    assert transforms.transform_dict({"a.+": None}, inp) == {}
    inp = {"a": {"b": 1, "c": 1}, "b": {"c": 2}}
    assert transforms.transform_dict({"(.+)/c": r"\1/d"}, inp) == {"a": {"b": 1, "d": 1}, "b": {"d": 2}}
    with pytest.raises(ValueError, match="aliases a node"):
        transforms.transform_dict({"a/b": "a"}, {"a": {"b": 1, "c": 2}})


def test_extract_prompt_from_task():
    t = transforms.PromptFromLeRobotTask({1: "Hello, world!"})
    assert t({"task_index": 1})["prompt"] == "Hello, world!"
    with pytest.raises(ValueError, match="task_index=2 not found in task mapping"):
        t({"task_index": 2})
    with pytest.raises(ValueError, match="task_index"):
        t({})


def test_tokenize_prompt_errors_and_wiring():
    sp_model = G["tok.model"].tobytes()
    tok = tokenizer.PaligemmaTokenizer(max_len=12, model=sp_model)
    data = transforms.TokenizePrompt(tok)({"prompt": "Hello, world!"})
    t, m = tok.tokenize("Hello, world!")
    assert np.array_equal(t, data["tokenized_prompt"]) and np.array_equal(m, data["tokenized_prompt_mask"])
    with pytest.raises(ValueError, match="Prompt is required"):
        transforms.TokenizePrompt(tok)({})
    with pytest.raises(ValueError, match="State is required"):
        transforms.TokenizePrompt(tok, discrete_state_input=True)({"prompt": "x"})
    assert transforms.TokenizePrompt(tok)({"prompt": np.asarray("fold")})["tokenized_prompt"].shape == (12,)  # 0-d array prompt
    with pytest.raises(FileNotFoundError):
        os.environ.pop("KAI0_PALIGEMMA_TOKENIZER", None)
        tokenizer.PaligemmaTokenizer()


def test_resize_with_pad_shapes():
    for shape, (h, w) in (((2, 10, 10, 3), (20, 20)), ((3, 30, 30, 3), (15, 15)), ((1, 50, 50, 3), (50, 50)), ((1, 256, 320, 3), (60, 80))):
        out = image_tools.resize_with_pad(np.zeros(shape, dtype=np.uint8), h, w)
        assert out.shape == (shape[0], h, w, 3) and np.all(out == 0)


MSG_CASES = [1, 1.0, "hello", np.bool_(True), np.array([1, 2, 3])[0], np.str_("asdf"), [1, 2, 3], {"key": "value"}, {"key": [1, 2, 3]},
             np.array(1.0), np.array([1, 2, 3], dtype=np.int32), np.array(["asdf", "qwer"]), np.array([True, False]),
             np.array([[1.0, 2.0], [3.0, 4.0]], dtype=np.float32), np.array([[[1, 2], [3, 4]], [[5, 6], [7, 8]]], dtype=np.int16),
             np.array([np.nan, np.inf, -np.inf]), {"arr": np.array([1, 2, 3]), "nested": {"arr": np.array([4, 5, 6])}},
             [np.array([1, 2]), np.array([3, 4])], np.zeros((3, 4, 5), dtype=np.float32), np.ones((2, 3), dtype=np.float64)]  # fmt: skip


def _same(a, b):
    if isinstance(a, dict):
        assert a.keys() == b.keys()
        for k in a:
            _same(a[k], b[k])
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    elif isinstance(a, np.ndarray):
        assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=a.dtype.kind == "f")
    else:
        assert a == b


@pytest.mark.parametrize("data", MSG_CASES, ids=range(len(MSG_CASES)))
def test_msgpack_pack_unpack(data):
    _same(data, msgpack_numpy.unpackb(msgpack_numpy.packb(data)))


def test_msgpack_refuses_object_arrays():
    with pytest.raises(ValueError, match="Unsupported dtype"):
        msgpack_numpy.packb(np.array([{"a": 1}], dtype=object))


# -------------------------------------------------------------------------------------------- 2. reference-executed vectors
def test_image_tools_against_reference_output():
    i = 0
    while f"img.{i}.in" in G:
        h, w = G[f"img.{i}.hw"]
        out = image_tools.resize_with_pad(G[f"img.{i}.in"], int(h), int(w))
        assert out.dtype == np.uint8 and np.array_equal(out, G[f"img.{i}.out"]), i
        i += 1
    assert i == 7
    assert np.array_equal(image_tools.convert_to_uint8(G["img.float.in"]), G["img.float.out"])
    same = np.zeros((4, 4, 3), dtype=np.uint8)
    assert image_tools.resize_with_pad(same, 4, 4) is same and image_tools.convert_to_uint8(same) is same


def test_msgpack_wire_bytes_match_the_reference():
    rng = np.random.default_rng(20240926)  # regenerate the generator's objects: same seed, same draw order up to MSG
    for shape in [(2, 10, 10, 3), (3, 30, 30, 3), (48, 64, 3), (1, 64, 40, 3), (2, 2, 37, 53, 3), (25, 25, 3), (1, 256, 320, 3)]:
        rng.integers(0, 256, shape, dtype=np.uint8)
    rng.random((5, 7, 3))
    msgs = [1, 1.5, "hello", np.bool_(True), np.int64(7), np.float32(2.5), [1, 2, 3], {"key": [1, 2, 3]}, np.array(1.0),
            np.array([1, 2, 3], dtype=np.int32), np.array(["asdf", "qwer"]), np.array([True, False]),
            np.array([[1.0, 2.0], [3.0, 4.0]], dtype=np.float32), np.array([np.nan, np.inf, -np.inf]),
            {"arr": np.arange(6, dtype=np.int16).reshape(2, 3), "nested": {"arr": np.array([4.0, 5.0])}},
            {"actions": rng.standard_normal((50, 14)).astype(np.float32), "policy_timing": {"infer_ms": 12.5}}]  # fmt: skip
    for i, obj in enumerate(msgs):
        ref = G[f"msg.{i}.bytes"].tobytes()
        assert msgpack_numpy.packb(obj) == ref, i          # same bytes on the wire
        _same(obj, msgpack_numpy.unpackb(ref))             # and the reference's bytes decode to the object


def _stats(prefix):
    return {k: NormStats(**{f: G[f"{prefix}.{k}.{f}"] for f in ("mean", "std", "q01", "q99")}) for k in ("state", "actions")}


def test_numeric_transforms_against_reference_output():
    stats = _stats("tf.stats")
    s14, s32, a14, a32 = G["tf.state14"], G["tf.state32"], G["tf.act14"], G["tf.act32"]
    for q in (False, True):
        n = transforms.Normalize(stats, use_quantiles=q)({"state": s14.copy(), "actions": a14.copy(), "other": np.ones(3)})
        assert np.array_equal(n["state"], G[f"tf.norm.q{int(q)}.state"]) and np.array_equal(n["actions"], G[f"tf.norm.q{int(q)}.actions"])
        assert np.array_equal(n["other"], np.ones(3))
        n = transforms.Normalize(stats, use_quantiles=q)({"state": s14[:9].copy()})
        assert np.array_equal(n["state"], G[f"tf.norm_short.q{int(q)}.state"])
        u = transforms.Unnormalize(stats, use_quantiles=q)({"state": s32.copy(), "actions": a32.copy()})
        assert np.array_equal(u["state"], G[f"tf.unnorm.q{int(q)}.state"]) and np.array_equal(u["actions"], G[f"tf.unnorm.q{int(q)}.actions"])
    with pytest.raises(ValueError, match="not found in tree"):
        transforms.Unnormalize(stats)({"state": s32})  # output side is always strict
    with pytest.raises(ValueError, match="not found in tree"):
        transforms.Normalize(stats, strict=True)({"state": s14})
    with pytest.raises(ValueError, match="missing q01 or q99"):
        transforms.Normalize({"state": NormStats(mean=np.zeros(2), std=np.ones(2))}, use_quantiles=True)
    assert transforms.Normalize(None)({"x": 1}) == {"x": 1}
    mask = transforms.make_bool_mask(6, -1, 6, -1)
    assert np.array_equal(np.asarray(mask), G["tf.mask"])
    d = transforms.DeltaActions(mask)({"state": s14.copy(), "actions": a14.copy()})
    assert np.array_equal(d["actions"], G["tf.delta.actions"])
    a = transforms.AbsoluteActions(mask)({"state": s14.copy(), "actions": d["actions"].copy()})
    assert np.array_equal(a["actions"], G["tf.absolute.actions"])
    p = transforms.PadStatesAndActions(32)({"state": s14.copy(), "actions": a14.copy()})
    assert np.array_equal(p["state"], G["tf.pad.state"]) and np.array_equal(p["actions"], G["tf.pad.actions"])
    assert np.array_equal(transforms.SubsampleActions(3)({"actions": a14.copy()})["actions"], G["tf.subsample.actions"])
    r = transforms.RepackTransform({"images": {"cam_high": "observation/images/top"}, "state": "observation/state", "actions": "action"})(
        {"observation": {"images": {"top": np.arange(4)}, "state": np.arange(3)}, "action": np.arange(5)})  # fmt: skip
    assert sorted(transforms.flatten_dict(r)) == list(G["tf.repack.keys"])
    out = transforms.ResizeImages(28, 28)({"image": {"cam": G["tf.resize.in"]}})["image"]["cam"]
    assert np.array_equal(out, G["tf.resize.out"])
    assert transforms.InsertAdvantageIntoPrompt()({"prompt": "fold the cloth", "advantage": 0.123456})["prompt"] == str(G["tf.adv.prompt"])
    g = transforms.Group(inputs=[1], outputs=[2]).push(inputs=[3], outputs=[4])
    assert tuple(g.inputs) == (1, 3) and tuple(g.outputs) == (4, 2)


def test_tokenizer_against_reference_output():
    sp_model = G["tok.model"].tobytes()
    state = G["tok.state"]
    for i, prompt in enumerate(G["tok.prompts"]):
        for max_len in (16, 48, 200):
            for with_state in (False, True):
                tok = tokenizer.PaligemmaTokenizer(max_len=max_len, model=sp_model)
                t, m = tok.tokenize(str(prompt), state if with_state else None)
                key = f"tok.{i}.{max_len}.{int(with_state)}"
                assert t.shape == (max_len,) and np.array_equal(t, G[key + ".tokens"]) and np.array_equal(m, G[key + ".mask"]), key
    b = tokenizer.discretize_state(np.array([-1.5, -1.0, -0.999, 0.0, 0.999, 1.0, 1.5]))
    assert list(b) == [-1, 0, 0, 128, 255, 255, 255]


def _cams():
    cams = {k[7:]: G[k] for k in G.files if k.startswith("ag.cam.")}
    cams["hand_right"] = torch.from_numpy(cams["hand_right"])  # one camera arrives as a torch tensor
    return cams


def test_agilex_transforms_against_reference_output():
    for mt in ("PI0", "PI05"):
        o = agilex_policy.AgilexInputs(action_dim=32, model_type=mt.lower())(
            {"images": _cams(), "state": G["ag.state"].copy(), "actions": G["ag.actions"].copy(), "prompt": "fold", "progress": np.float32(0.25)})  # fmt: skip
        flat = transforms.flatten_dict(o)
        assert sorted(flat) == list(G[f"ag.{mt}.keys"])
        for k, v in flat.items():
            ref = G[f"ag.{mt}.{k}"]
            assert np.array_equal(np.asarray(v), ref) and np.asarray(v).dtype == ref.dtype, (mt, k)
    assert flat["state"][3] == 0 and flat["state"][9] == 0 and flat["actions"][5, 2] == 0  # glitches beyond +-pi are zeroed
    assert np.array_equal(agilex_policy.AgilexOutputs()({"actions": G["tf.act32"]})["actions"], G["ag.out.actions"])
    with pytest.raises(ValueError, match="Expected images to contain"):
        agilex_policy.AgilexInputs(action_dim=32)({"images": {"webcam": np.zeros((3, 4, 4))}, "state": np.zeros(14)})
    with pytest.raises(ValueError, match="Camera hand_right not found"):
        cams = _cams()
        del cams["hand_right"]
        agilex_policy.AgilexInputs(action_dim=32)({"images": cams, "state": np.zeros(14)})


class _FakeModel(torch.nn.Module):
    """Same fake as the generator's: every Observation field leaves a trace in the chunk."""

    def sample_actions(self, device, observation, noise=None, num_steps=10):
        b = observation.state.shape[0]
        base = observation.state.to(torch.float32)[:, None, :].expand(b, 50, 32).clone()
        base += sum(v.to(torch.float32).mean() for v in observation.images.values())
        base += observation.tokenized_prompt.to(torch.float32).sum() * 1e-3 + observation.tokenized_prompt_mask.sum() * 1e-2
        return base + (noise.to(torch.float32) if noise is not None else 0) + num_steps


def test_policy_infer_flow_against_reference_output():
    stats = _stats("pol.stats")
    tok = tokenizer.PaligemmaTokenizer(max_len=48, model=G["tok.model"].tobytes())
    pol = policy.create_policy(_FakeModel(), norm_stats=stats, tokenizer=tok, action_dim=32, use_quantile_norm=True, image_size=28,
                               robot_inputs=[agilex_policy.AgilexInputs(action_dim=32, model_type="pi05")],
                               robot_outputs=[agilex_policy.AgilexOutputs()], sample_kwargs={"num_steps": 10},
                               metadata={"robot": "agilex"}, pytorch_device="cpu")  # fmt: skip
    cams = {k: v for k, v in _cams().items() if not k.startswith("his_")}
    obs = {"images": cams, "state": G["ag.state"].copy(), "prompt": "fold the cloth"}
    res = pol.infer({**obs, "images": dict(cams)})
    assert sorted(res) == list(G["pol.plain.keys"]) and res["actions"].shape == (50, 14)
    assert np.array_equal(res["actions"], G["pol.plain.actions"])
    res = pol.infer({**obs, "images": dict(cams)}, noise=G["pol.noise"])
    assert np.array_equal(res["actions"], G["pol.noise.actions"])
    assert res["policy_timing"]["infer_ms"] >= 0 and pol.metadata == {"robot": "agilex"}
    assert "prompt" in obs and obs["state"][3] == 4.0  # the caller's dict is not modified
    # round 5: the frames' resize behind the host-to-device copy (kai0_amd.device_resize; on by default when the model sits on a GPU,
    # forced here): the reference's output, bit for bit
    pol2 = policy.create_policy(_FakeModel(), norm_stats=stats, tokenizer=tok, action_dim=32, use_quantile_norm=True, image_size=28,
                                robot_inputs=[agilex_policy.AgilexInputs(action_dim=32, model_type="pi05")],
                                robot_outputs=[agilex_policy.AgilexOutputs()], sample_kwargs={"num_steps": 10},
                                metadata={"robot": "agilex"}, pytorch_device="cpu", device_resize="force")  # fmt: skip
    assert pol2._device_resize is not None and pol._device_resize is None
    assert np.array_equal(pol2.infer({**obs, "images": dict(cams)})["actions"], G["pol.plain.actions"])
    assert np.array_equal(pol2.infer({**obs, "images": dict(cams)}, noise=G["pol.noise"])["actions"], G["pol.noise.actions"])


@pytest.mark.parametrize("h0,w0,H,W", [(480, 640, 224, 224), (224, 224, 224, 224), (100, 120, 224, 224), (720, 1280, 224, 224),
                                       (481, 643, 224, 224), (300, 200, 224, 224), (37, 53, 64, 48), (50, 1000, 224, 224)])
def test_device_resize_is_pillow_bit_for_bit(h0, w0, H, W):
    """kai0_amd.device_resize (torch integer arithmetic, any device) against image_tools.resize_with_pad (Pillow BILINEAR, what the
    reference's serve path runs, openpi_client/image_tools.py:7-58): noise, saturated and ramp frames, down- and upscaling, odd sizes,
    a batch with leading dimensions."""
    from kai0_amd import device_resize

    rng = np.random.default_rng(h0 * 1000 + w0)
    frames = [rng.integers(0, 256, (h0, w0, 3), dtype=np.uint8), np.full((h0, w0, 3), 255, np.uint8),
              (np.indices((h0, w0)).sum(0) % 256).astype(np.uint8)[..., None].repeat(3, -1)]
    for f in frames:
        assert np.array_equal(device_resize.resize_with_pad_u8(torch.from_numpy(f), H, W).numpy(), image_tools.resize_with_pad(f, H, W))
    batch = np.stack(frames)[None]
    assert np.array_equal(device_resize.resize_with_pad_u8(torch.from_numpy(batch), H, W).numpy(), image_tools.resize_with_pad(batch, H, W))
    with pytest.raises(TypeError):
        device_resize.resize_with_pad_u8(torch.zeros(4, 4, 3), 2, 2)
