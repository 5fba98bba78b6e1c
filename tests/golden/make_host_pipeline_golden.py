"""Golden vectors for the host side of the serve / data path, produced by EXECUTING THE REFERENCE'S OWN CODE
in the build container (needs /root/reference; nothing is copied, the source is read at run time):

  * `openpi_client.image_tools` and `openpi_client.msgpack_numpy` import as they are (numpy + PIL + msgpack only);
  * `openpi.transforms`, `openpi.policies.agilex_policy`, `openpi.policies.policy.Policy` and
    `openpi.models.tokenizer.PaligemmaTokenizer.tokenize` are lifted out of their files with `ast` and run against stubs for
    the un-vendored imports (flax.traverse_util flatten / unflatten with '/', jax.tree.map over dicts) and for the pieces that
    need the network (the sentencepiece model: a tiny one is trained here and stored in the fixture).

    python tests/golden/make_host_pipeline_golden.py   ->  tests/golden/host_pipeline.npz

tests/test_host_pipeline_cpu.py replays the same inputs through kai0_amd.{transforms,image_tools,msgpack_numpy,tokenizer,
agilex_policy,policy} and requires identical results."""
import ast
import dataclasses
import enum
import io
import logging
import os
import re
import sys
import time
import types
import typing

import numpy as np
import sentencepiece
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, f"{REF}/packages/openpi-client/src")
from openpi_client import image_tools as ref_image_tools  # noqa: E402
from openpi_client import msgpack_numpy as ref_msgpack  # noqa: E402

OUT = {}
rng = np.random.default_rng(20240926)


def put(name, v):
    OUT[name] = np.asarray(v)


def lift(path, names, ns):
    tree = ast.parse(open(path).read())
    found = set()
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
            found.add(node.name)
    assert found == set(names), set(names) - found
    return ns


def lift_method(path, cls, name, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    sub.decorator_list = []
                    exec(compile(ast.Module([sub], []), path, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


# ---- stand-ins for un-vendored third parties --------------------------------------------------------------------------------
def _flatten(tree, sep="/"):
    out = {}

    def rec(node, pre):
        if isinstance(node, dict) and node:
            for k, v in node.items():
                rec(v, pre + [str(k)])
        else:
            out[sep.join(pre)] = node

    rec(tree, [])
    return out


def _unflatten(flat, sep="/"):
    out = {}
    for k, v in flat.items():
        parts = k.split(sep)
        d = out
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return out


def _tree_map(fn, tree):
    return {k: _tree_map(fn, v) for k, v in tree.items()} if isinstance(tree, dict) else fn(tree)


class _Sub:  # `at.PyTree[str]` in annotations
    def __getitem__(self, item):
        return typing.Any


jax_stub = types.SimpleNamespace(tree=types.SimpleNamespace(map=_tree_map))
NormStats = dataclasses.make_dataclass("NormStats", [("mean", typing.Any), ("std", typing.Any), ("q01", typing.Any, None), ("q99", typing.Any, None)])  # fmt: skip
T = types.SimpleNamespace  # noqa: N816
tns = {"dataclasses": dataclasses, "np": np, "re": re, "jax": jax_stub, "image_tools": ref_image_tools,
       "traverse_util": T(flatten_dict=_flatten, unflatten_dict=_unflatten), "_tokenizer": T(PaligemmaTokenizer=typing.Any),
       "at": T(PyTree=_Sub()), "_normalize": T(NormStats=NormStats), "NormStats": NormStats, "DataDict": typing.Any,
       "Callable": typing.Callable, "Mapping": typing.Mapping, "Sequence": typing.Sequence, "T": typing.Any, "S": typing.Any,
       "DataTransformFn": object}  # fmt: skip
lift(f"{REF}/src/openpi/transforms.py",
     ["Group", "CompositeTransform", "compose", "RepackTransform", "InjectDefaultPrompt", "InsertAdvantageIntoPrompt", "Normalize",
      "Unnormalize", "ResizeImages", "SubsampleActions", "DeltaActions", "AbsoluteActions", "TokenizePrompt",
      "PromptFromLeRobotTask", "PadStatesAndActions", "flatten_dict", "unflatten_dict", "transform_dict", "apply_tree",
      "pad_to_dim", "make_bool_mask", "_assert_quantile_stats"], tns)  # fmt: skip
RT = T(**tns)

# ---- 1. image_tools -----------------------------------------------------------------------------------------------------------
for i, (shape, hw) in enumerate([((2, 10, 10, 3), (20, 20)), ((3, 30, 30, 3), (15, 15)), ((48, 64, 3), (28, 28)), ((1, 64, 40, 3), (28, 28)),
                                 ((2, 2, 37, 53, 3), (32, 24)), ((25, 25, 3), (25, 25)), ((1, 256, 320, 3), (60, 80))]):  # fmt: skip
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    put(f"img.{i}.in", img)
    put(f"img.{i}.hw", hw)
    put(f"img.{i}.out", ref_image_tools.resize_with_pad(img, *hw))
f = rng.random((5, 7, 3)).astype(np.float32)
put("img.float.in", f)
put("img.float.out", ref_image_tools.convert_to_uint8(f))

# ---- 2. msgpack_numpy -----------------------------------------------------------------------------------------------------------
MSG = [1, 1.5, "hello", np.bool_(True), np.int64(7), np.float32(2.5), [1, 2, 3], {"key": [1, 2, 3]}, np.array(1.0),
       np.array([1, 2, 3], dtype=np.int32), np.array(["asdf", "qwer"]), np.array([True, False]),
       np.array([[1.0, 2.0], [3.0, 4.0]], dtype=np.float32), np.array([np.nan, np.inf, -np.inf]),
       {"arr": np.arange(6, dtype=np.int16).reshape(2, 3), "nested": {"arr": np.array([4.0, 5.0])}},
       {"actions": rng.standard_normal((50, 14)).astype(np.float32), "policy_timing": {"infer_ms": 12.5}}]  # fmt: skip
for i, obj in enumerate(MSG):
    put(f"msg.{i}.bytes", np.frombuffer(ref_msgpack.packb(obj), dtype=np.uint8))

# ---- 3. transforms ----------------------------------------------------------------------------------------------------------------
stats = {"state": NormStats(mean=rng.standard_normal(14), std=rng.random(14) + 0.1, q01=-rng.random(14) - 0.5, q99=rng.random(14) + 0.5),
         "actions": NormStats(mean=rng.standard_normal(14), std=rng.random(14) + 0.1, q01=-rng.random(14) - 0.5, q99=rng.random(14) + 0.5)}  # fmt: skip
for k, v in stats.items():
    for fld in ("mean", "std", "q01", "q99"):
        put(f"tf.stats.{k}.{fld}", getattr(v, fld))
state14, state32 = rng.standard_normal(14), rng.standard_normal(32)
act14, act32 = rng.standard_normal((50, 14)), rng.standard_normal((50, 32))
put("tf.state14", state14), put("tf.state32", state32), put("tf.act14", act14), put("tf.act32", act32)
for q in (False, True):
    n = RT.Normalize(stats, use_quantiles=q)({"state": state14.copy(), "actions": act14.copy(), "other": np.ones(3)})
    put(f"tf.norm.q{int(q)}.state", n["state"]), put(f"tf.norm.q{int(q)}.actions", n["actions"])
    n = RT.Normalize(stats, use_quantiles=q)({"state": state14[:9].copy()})  # statistics longer than the vector
    put(f"tf.norm_short.q{int(q)}.state", n["state"])
    u = RT.Unnormalize(stats, use_quantiles=q)({"state": state32.copy(), "actions": act32.copy()})  # shorter than the vector
    put(f"tf.unnorm.q{int(q)}.state", u["state"]), put(f"tf.unnorm.q{int(q)}.actions", u["actions"])
mask = RT.make_bool_mask(6, -1, 6, -1)
put("tf.mask", mask)
d = RT.DeltaActions(mask)({"state": state14.copy(), "actions": act14.copy()})
put("tf.delta.actions", d["actions"])
a = RT.AbsoluteActions(mask)({"state": state14.copy(), "actions": d["actions"].copy()})
put("tf.absolute.actions", a["actions"])
p = RT.PadStatesAndActions(32)({"state": state14.copy(), "actions": act14.copy()})
put("tf.pad.state", p["state"]), put("tf.pad.actions", p["actions"])
put("tf.subsample.actions", RT.SubsampleActions(3)({"actions": act14.copy()})["actions"])
r = RT.RepackTransform({"images": {"cam_high": "observation/images/top"}, "state": "observation/state", "actions": "action"})(
    {"observation": {"images": {"top": np.arange(4)}, "state": np.arange(3)}, "action": np.arange(5)})  # fmt: skip
put("tf.repack.keys", sorted(RT.flatten_dict(r)))
img = {"cam": rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)}
put("tf.resize.in", img["cam"]), put("tf.resize.out", RT.ResizeImages(28, 28)({"image": dict(img)})["image"]["cam"])
put("tf.adv.prompt", RT.InsertAdvantageIntoPrompt()({"prompt": "fold the cloth", "advantage": 0.123456})["prompt"])

# ---- 4. tokenizer -------------------------------------------------------------------------------------------------------------------
corpus = ["Task: fold the cloth, State: 1 2 3 4 5 6 7 8 9 10;", "Action: pick up the red block and place it in the bin",
          "flatten and fold the t-shirt on the table 0 11 22 33 44 55 66 77 88 99 100 128 255 200 17"] * 50  # fmt: skip
w = io.BytesIO()
sentencepiece.SentencePieceTrainer.train(sentence_iterator=iter(corpus), model_writer=w, vocab_size=120, model_type="bpe", bos_id=2,
                                         eos_id=1, unk_id=3, pad_id=0, character_coverage=1.0, hard_vocab_limit=False,
                                         normalization_rule_name="identity", user_defined_symbols=["\n"], minloglevel=2)  # fmt: skip
put("tok.model", np.frombuffer(w.getvalue(), dtype=np.uint8))
sp = sentencepiece.SentencePieceProcessor(model_proto=w.getvalue())
ref_tokenize = lift_method(f"{REF}/src/openpi/models/tokenizer.py", "PaligemmaTokenizer", "tokenize", {"np": np, "logging": logging})
tok_state = np.concatenate([rng.uniform(-1.2, 1.2, 12), [-1.0, 1.0]])
put("tok.state", tok_state)
PROMPTS = ["fold the cloth", "  pick_up the\nred block ", "flatten and fold the t-shirt on the table and place it in the bin " * 3]
for i, prompt in enumerate(PROMPTS):
    for max_len in (16, 48, 200):
        for with_state in (False, True):
            self_ = T(_tokenizer=sp, _max_len=max_len)
            toks, msk = ref_tokenize(self_, prompt, tok_state if with_state else None)
            put(f"tok.{i}.{max_len}.{int(with_state)}.tokens", toks), put(f"tok.{i}.{max_len}.{int(with_state)}.mask", msk)
OUT["tok.prompts"] = np.asarray(PROMPTS)

# ---- 5. Agilex robot transforms -----------------------------------------------------------------------------------------------------
ModelType = enum.Enum("ModelType", {"PI0": "pi0", "PI0_FAST": "pi0_fast", "PI05": "pi05", "PI0_RTC": "pi0_rtc", "PI05_RTC": "pi05_rtc"})
ans = {"dataclasses": dataclasses, "np": np, "torch": torch, "ClassVar": typing.ClassVar, "transforms": RT,
       "_model": T(ModelType=ModelType)}  # fmt: skip
lift(f"{REF}/src/openpi/policies/agilex_policy.py", ["AgilexInputs", "AgilexOutputs"], ans)
cams = {"top_head": rng.random((3, 24, 32)).astype(np.float32), "hand_left": rng.integers(0, 256, (24, 32, 3), dtype=np.uint8),
        "hand_right": torch.from_numpy(rng.random((3, 24, 32)).astype(np.float32)),
        "his_-100_top_head": rng.random((3, 24, 32)).astype(np.float32)}  # fmt: skip
ag_state = rng.uniform(-2, 2, 14)
ag_state[3], ag_state[9] = 4.0, -3.5  # glitches beyond +-pi
ag_actions = rng.uniform(-2, 2, (50, 14))
ag_actions[5, 2] = 3.3
for k, v in cams.items():
    put(f"ag.cam.{k}", v.numpy() if isinstance(v, torch.Tensor) else v)
put("ag.state", ag_state), put("ag.actions", ag_actions)
for mt in ("PI0", "PI05"):
    o = ans["AgilexInputs"](action_dim=32, model_type=ModelType[mt])(
        {"images": dict(cams), "state": ag_state.copy(), "actions": ag_actions.copy(), "prompt": "fold", "progress": np.float32(0.25)})  # fmt: skip
    for k, v in _flatten(o).items():
        put(f"ag.{mt}.{k}", v)
    OUT[f"ag.{mt}.keys"] = np.asarray(sorted(_flatten(o)))
put("ag.out.actions", ans["AgilexOutputs"]()({"actions": act32.copy()})["actions"])

# ---- 6. Policy.infer flow (fake model; transforms and batching as in the reference) -----------------------------------------------------
from kai0_amd.preprocessing import Observation  # noqa: E402  (the reference's Observation.from_dict pulls in jax types)

pns = {"np": np, "torch": torch, "time": time, "jax": jax_stub, "jnp": None, "_transforms": RT, "_model": T(Observation=Observation, BaseModel=typing.Any),
       "nnx_utils": None, "Sequence": typing.Sequence, "Any": typing.Any, "at": T(KeyArrayLike=typing.Any), "_transforms_DataTransformFn": None, "override": lambda f: f}  # fmt: skip
ref_init = lift_method(f"{REF}/src/openpi/policies/policy.py", "Policy", "__init__", pns)
ref_infer = lift_method(f"{REF}/src/openpi/policies/policy.py", "Policy", "infer", pns)


class FakeModel(torch.nn.Module):
    """actions = f(state, images, tokens, noise): every input of the Observation leaves a trace in the output"""

    def sample_actions(self, device, observation, noise=None, num_steps=10):
        b = observation.state.shape[0]
        base = observation.state.to(torch.float32)[:, None, :].expand(b, 50, 32).clone()
        base += sum(v.to(torch.float32).mean() for v in observation.images.values())
        base += observation.tokenized_prompt.to(torch.float32).sum() * 1e-3 + observation.tokenized_prompt_mask.sum() * 1e-2
        return base + (noise.to(torch.float32) if noise is not None else 0) + num_steps


robot_obs = {"images": {k: v for k, v in cams.items() if not k.startswith("his_")}, "state": ag_state.copy(), "prompt": "fold the cloth"}
tok48 = T(tokenize=lambda prompt, state=None: ref_tokenize(T(_tokenizer=sp, _max_len=48), prompt, state))
# in the real pipeline the statistics are computed AFTER the robot transform padded state / actions to the model width
stats32 = {k: NormStats(mean=rng.standard_normal(32), std=rng.random(32) + 0.1, q01=-rng.random(32) - 0.5, q99=rng.random(32) + 0.5)
           for k in ("state", "actions")}  # fmt: skip
for k, v in stats32.items():
    for fld in ("mean", "std", "q01", "q99"):
        put(f"pol.stats.{k}.{fld}", getattr(v, fld))
chain_in = [RT.InjectDefaultPrompt(None), ans["AgilexInputs"](action_dim=32, model_type=ModelType.PI05), RT.Normalize(stats32, use_quantiles=True),
            RT.InjectDefaultPrompt(None), RT.ResizeImages(28, 28), RT.TokenizePrompt(tok48, discrete_state_input=True), RT.PadStatesAndActions(32)]  # fmt: skip
chain_out = [RT.Unnormalize(stats32, use_quantiles=True), ans["AgilexOutputs"]()]
pol = T()
ref_init(pol, FakeModel(), transforms=chain_in, output_transforms=chain_out, sample_kwargs={"num_steps": 10}, metadata={"robot": "agilex"},
         pytorch_device="cpu", is_pytorch=True)  # fmt: skip
pol_noise = rng.standard_normal((50, 32)).astype(np.float32)
put("pol.noise", pol_noise)
for tag, kw in (("plain", {}), ("noise", {"noise": pol_noise})):
    res = ref_infer(pol, {**robot_obs, "images": dict(robot_obs["images"])}, **kw)
    put(f"pol.{tag}.actions", res["actions"])
    OUT[f"pol.{tag}.keys"] = np.asarray(sorted(res))
    assert "infer_ms" in res["policy_timing"]

np.savez_compressed(os.path.join(HERE, "host_pipeline.npz"), **OUT)
print("wrote host_pipeline.npz:", len(OUT), "arrays,", os.path.getsize(os.path.join(HERE, "host_pipeline.npz")), "bytes")
