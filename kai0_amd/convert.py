"""JAX (Flax / NNX) parameter tree <-> the PyTorch state-dict of the hot path.

kai0 ships JAX checkpoints; the torch-protocol model (and therefore this framework) needs the same weights under the
state-dict contract of SURVEY.md §8 a16.  The fork carries no converter, so the mapping below is derived from the two module
trees themselves — each rule cites the JAX definition whose einsum / matmul fixes the axis order:

  PaliGemma/llm (src/openpi/models/gemma.py; expert parameters carry the suffix "_1", `_name` :443-446; every per-layer tensor
  has a leading [depth] axis from nn.scan :366-381)
    embedder/input_embedding [V, D]                      -> language_model.embed_tokens.weight (lm_head is tied)   :142-146
    layers/attn/q_einsum/w [L, N, D, H]  "BTD,NDH->BTNH" -> q_proj.weight [N*H, D]                                 :185-191
    layers/attn/kv_einsum/w [L, 2, K, D, H]  "BSD,2KDH->2BSKH" -> k_proj / v_proj.weight [K*H, D]                  :192-199
    layers/attn/attn_vec_einsum/w [L, N, H, D]  "BTNH,NHD->BTD" -> o_proj.weight [D, N*H]                          :238-245
    layers/mlp/gating_einsum [L, 2, D, F]  dot(x, w[0]) = gate, dot(x, w[1]) = up -> gate_proj / up_proj.weight [F, D]  :262-270
    layers/mlp/linear [L, F, D]  dot(a, w)              -> down_proj.weight [D, F]                                 :273-278
    layers/pre_attention_norm/scale [L, D], pre_ffw_norm/scale, final_norm/scale   -> *.weight ((1 + w) on both sides) :121-125
    layers/pre_attention_norm_1/Dense_0/{kernel [L, D, 3D], bias [L, 3D]} (adaRMS) -> *.dense.{weight [3D, D], bias}   :128
  PaliGemma/img (src/openpi/models/siglip.py, big_vision ViT; encoder blocks scanned under "encoderblock" :133-141)
    embedding/{kernel [p, p, 3, W], bias}                 -> embeddings.patch_embedding.{weight [W, 3, p, p], bias}   :216-222
    pos_embedding [1, n, W]                               -> embeddings.position_embedding.weight [n, W]             :229
    Transformer/encoderblock/LayerNorm_{0,1}/{scale, bias}-> layer_norm{1,2}.{weight, bias}                          :87,98
    .../MultiHeadDotProductAttention_0/{query,key,value}/{kernel [L, W, h, d], bias [L, h, d]}
                                                          -> self_attn.{q,k,v}_proj.{weight [h*d, W], bias [h*d]}    (flax DenseGeneral)
    .../MultiHeadDotProductAttention_0/out/{kernel [L, h, d, W], bias [L, W]} -> self_attn.out_proj.{weight [W, h*d], bias}
    .../MlpBlock_0/Dense_{0,1}/{kernel, bias}             -> mlp.fc{1,2}.{weight = kernel^T, bias}                   :69-72
    Transformer/encoder_norm/{scale, bias}                -> post_layernorm.{weight, bias}                            :161
    head/{kernel [W, D], bias}                            -> multi_modal_projector.linear.{weight [D, W], bias}       :286
  top level nnx.Linear (src/openpi/models/pi0.py:93-101): {action_in_proj, action_out_proj, time_mlp_in, time_mlp_out}/
    {kernel [in, out], bias}                              -> *.{weight = kernel^T, bias}

No entry exists on the JAX side for `gemma_expert.lm_head.weight` (dead weight of the torch module tree);
`fill_missing=True` zero-fills it so that `load_state_dict(strict=True)` works.  RoPE needs no permutation: both sides
rotate the two halves of the head dimension (`_apply_rope` gemma.py:422-434, `rotate_half` modeling_gemma.py:149-153).

Arrays are numpy on the JAX side (what `restore_params(restore_type=np.ndarray)` returns, flattened with '/'), torch tensors
on the other; dtypes are preserved, `to_bfloat16_for_selected_params` decides the storage dtypes afterwards as in the
reference.  tests/test_convert_cpu.py checks the axis conventions against the einsum forms quoted above."""

from __future__ import annotations

from collections.abc import Mapping

import numpy as np
import torch

PWE = "paligemma_with_expert."
LM = PWE + "paligemma.model.language_model."
EX = PWE + "gemma_expert.model."
VT = PWE + "paligemma.model.vision_tower.vision_model."
PROJ = PWE + "paligemma.model.multi_modal_projector.linear."
HEADS = ("action_in_proj", "action_out_proj", "time_mlp_in", "time_mlp_out")


def flatten_params(tree: Mapping, sep: str = "/") -> dict:
    """Nested param dict -> {'a/b/c': array}; strips a leading 'params' level and the trailing 'value' that `nnx.State`
    adds (model.py:360-364)."""
    flat = {}

    def rec(node, path):
        if isinstance(node, Mapping):
            for k, v in node.items():
                rec(v, path + (str(k),))
        else:
            flat[path] = node

    rec(tree, ())
    if flat and all(p[-1] == "value" for p in flat):
        flat = {p[:-1]: v for p, v in flat.items()}
    if flat and all(p[0] == "params" for p in flat):
        flat = {p[1:]: v for p, v in flat.items()}
    return {sep.join(p): v for p, v in flat.items()}


def _t(x) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x)))


def _np(x: torch.Tensor) -> np.ndarray:
    if x.dtype == torch.bfloat16:  # numpy has no bf16: widen (exact)
        x = x.to(torch.float32)
    return x.detach().cpu().numpy()


def jax_to_torch(params: Mapping, *, fill_missing: bool = False, vocab_size: int | None = None) -> dict[str, torch.Tensor]:
    """JAX params (nested or '/'-flattened) -> state-dict of kai0_amd.model.PI0Pytorch / the reference's PI0Pytorch."""
    p = {k: np.asarray(v) for k, v in (params if all(isinstance(k, str) and not isinstance(v, Mapping) for k, v in params.items())
                                       else flatten_params(params)).items()}  # fmt: skip
    sd: dict[str, torch.Tensor] = {}
    llm, img = "PaliGemma/llm/", "PaliGemma/img/"

    emb = p[llm + "embedder/input_embedding"]
    sd[LM + "embed_tokens.weight"] = _t(emb)
    sd[PWE + "paligemma.lm_head.weight"] = sd[LM + "embed_tokens.weight"]  # tied

    for sfx, dst, ada in (("", LM, False), ("_1", EX, True)):
        q = p[f"{llm}layers/attn/q_einsum{sfx}/w"]            # [L, N, D, H]
        kv = p[f"{llm}layers/attn/kv_einsum{sfx}/w"]          # [L, 2, K, D, H]
        o = p[f"{llm}layers/attn/attn_vec_einsum{sfx}/w"]     # [L, N, H, D]
        gating = p[f"{llm}layers/mlp{sfx}/gating_einsum"]     # [L, 2, D, F]
        linear = p[f"{llm}layers/mlp{sfx}/linear"]            # [L, F, D]
        L, N, D, H = q.shape
        K = kv.shape[2]
        for i in range(L):
            pre = f"{dst}layers.{i}."
            sd[pre + "self_attn.q_proj.weight"] = _t(q[i].transpose(0, 2, 1).reshape(N * H, D))
            sd[pre + "self_attn.k_proj.weight"] = _t(kv[i, 0].transpose(0, 2, 1).reshape(K * H, D))
            sd[pre + "self_attn.v_proj.weight"] = _t(kv[i, 1].transpose(0, 2, 1).reshape(K * H, D))
            sd[pre + "self_attn.o_proj.weight"] = _t(o[i].reshape(N * H, D).T)
            sd[pre + "mlp.gate_proj.weight"] = _t(gating[i, 0].T)
            sd[pre + "mlp.up_proj.weight"] = _t(gating[i, 1].T)
            sd[pre + "mlp.down_proj.weight"] = _t(linear[i].T)
            for jn, tn in (("pre_attention_norm", "input_layernorm"), ("pre_ffw_norm", "post_attention_layernorm")):
                if ada:
                    sd[f"{pre}{tn}.dense.weight"] = _t(p[f"{llm}layers/{jn}{sfx}/Dense_0/kernel"][i].T)
                    sd[f"{pre}{tn}.dense.bias"] = _t(p[f"{llm}layers/{jn}{sfx}/Dense_0/bias"][i])
                else:
                    sd[f"{pre}{tn}.weight"] = _t(p[f"{llm}layers/{jn}{sfx}/scale"][i])
        if ada:
            sd[dst + "norm.dense.weight"] = _t(p[f"{llm}final_norm{sfx}/Dense_0/kernel"].T)
            sd[dst + "norm.dense.bias"] = _t(p[f"{llm}final_norm{sfx}/Dense_0/bias"])
        else:
            sd[dst + "norm.weight"] = _t(p[f"{llm}final_norm{sfx}/scale"])

    ker = p[img + "embedding/kernel"]                          # [ph, pw, 3, W]
    sd[VT + "embeddings.patch_embedding.weight"] = _t(ker.transpose(3, 2, 0, 1))
    sd[VT + "embeddings.patch_embedding.bias"] = _t(p[img + "embedding/bias"])
    sd[VT + "embeddings.position_embedding.weight"] = _t(p[img + "pos_embedding"][0])
    blk = img + "Transformer/encoderblock/"
    att = blk + "MultiHeadDotProductAttention_0/"
    depth = p[blk + "LayerNorm_0/scale"].shape[0]
    for i in range(depth):
        pre = f"{VT}encoder.layers.{i}."
        for j, name in ((0, "layer_norm1"), (1, "layer_norm2")):
            sd[f"{pre}{name}.weight"] = _t(p[f"{blk}LayerNorm_{j}/scale"][i])
            sd[f"{pre}{name}.bias"] = _t(p[f"{blk}LayerNorm_{j}/bias"][i])
        for jn, tn in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj")):
            k = p[f"{att}{jn}/kernel"][i]                      # [W, h, d]
            sd[f"{pre}self_attn.{tn}.weight"] = _t(k.reshape(k.shape[0], -1).T)
            sd[f"{pre}self_attn.{tn}.bias"] = _t(p[f"{att}{jn}/bias"][i].reshape(-1))
        k = p[att + "out/kernel"][i]                           # [h, d, W]
        sd[pre + "self_attn.out_proj.weight"] = _t(k.reshape(-1, k.shape[-1]).T)
        sd[pre + "self_attn.out_proj.bias"] = _t(p[att + "out/bias"][i])
        for j, name in ((0, "fc1"), (1, "fc2")):
            sd[f"{pre}mlp.{name}.weight"] = _t(p[f"{blk}MlpBlock_0/Dense_{j}/kernel"][i].T)
            sd[f"{pre}mlp.{name}.bias"] = _t(p[f"{blk}MlpBlock_0/Dense_{j}/bias"][i])
    sd[VT + "post_layernorm.weight"] = _t(p[img + "Transformer/encoder_norm/scale"])
    sd[VT + "post_layernorm.bias"] = _t(p[img + "Transformer/encoder_norm/bias"])
    sd[PROJ + "weight"] = _t(p[img + "head/kernel"].T)
    sd[PROJ + "bias"] = _t(p[img + "head/bias"])

    for name in HEADS:
        if f"{name}/kernel" in p:
            sd[name + ".weight"] = _t(p[f"{name}/kernel"].T)
            sd[name + ".bias"] = _t(p[f"{name}/bias"])
    if fill_missing:
        width = p[f"{llm}layers/attn/q_einsum_1/w"].shape[2]
        sd[PWE + "gemma_expert.lm_head.weight"] = torch.zeros((vocab_size or emb.shape[0], width), dtype=sd[EX + "norm.dense.bias"].dtype)
    return sd


def torch_to_jax(sd: Mapping[str, torch.Tensor], *, num_heads: int = 8, num_kv_heads: int = 1,
                 siglip_heads: int = 16) -> dict[str, np.ndarray]:  # fmt: skip
    """The inverse map ('/'-flattened JAX tree, numpy; bf16 tensors are widened to f32 because numpy has no bf16)."""
    out: dict[str, np.ndarray] = {}
    llm, img = "PaliGemma/llm/", "PaliGemma/img/"
    out[llm + "embedder/input_embedding"] = _np(sd[LM + "embed_tokens.weight"])

    def depth_of(prefix):
        return 1 + max(int(k[len(prefix) :].split(".")[1]) for k in sd if k.startswith(prefix + "layers."))

    for sfx, src, ada in (("", LM, False), ("_1", EX, True)):
        L = depth_of(src)
        g = lambda i, name: _np(sd[f"{src}layers.{i}.{name}"])  # noqa: E731
        D = g(0, "self_attn.q_proj.weight").shape[1]
        H = g(0, "self_attn.q_proj.weight").shape[0] // num_heads
        N, K = num_heads, num_kv_heads
        out[f"{llm}layers/attn/q_einsum{sfx}/w"] = np.stack([g(i, "self_attn.q_proj.weight").reshape(N, H, D).transpose(0, 2, 1) for i in range(L)])
        out[f"{llm}layers/attn/kv_einsum{sfx}/w"] = np.stack(
            [np.stack([g(i, f"self_attn.{n}_proj.weight").reshape(K, H, D).transpose(0, 2, 1) for n in ("k", "v")]) for i in range(L)])  # fmt: skip
        out[f"{llm}layers/attn/attn_vec_einsum{sfx}/w"] = np.stack([g(i, "self_attn.o_proj.weight").T.reshape(N, H, D) for i in range(L)])
        out[f"{llm}layers/mlp{sfx}/gating_einsum"] = np.stack([np.stack([g(i, "mlp.gate_proj.weight").T, g(i, "mlp.up_proj.weight").T]) for i in range(L)])
        out[f"{llm}layers/mlp{sfx}/linear"] = np.stack([g(i, "mlp.down_proj.weight").T for i in range(L)])
        for jn, tn in (("pre_attention_norm", "input_layernorm"), ("pre_ffw_norm", "post_attention_layernorm")):
            if ada:
                out[f"{llm}layers/{jn}{sfx}/Dense_0/kernel"] = np.stack([g(i, tn + ".dense.weight").T for i in range(L)])
                out[f"{llm}layers/{jn}{sfx}/Dense_0/bias"] = np.stack([g(i, tn + ".dense.bias") for i in range(L)])
            else:
                out[f"{llm}layers/{jn}{sfx}/scale"] = np.stack([g(i, tn + ".weight") for i in range(L)])
        if ada:
            out[f"{llm}final_norm{sfx}/Dense_0/kernel"] = _np(sd[src + "norm.dense.weight"]).T
            out[f"{llm}final_norm{sfx}/Dense_0/bias"] = _np(sd[src + "norm.dense.bias"])
        else:
            out[f"{llm}final_norm{sfx}/scale"] = _np(sd[src + "norm.weight"])

    out[img + "embedding/kernel"] = _np(sd[VT + "embeddings.patch_embedding.weight"]).transpose(2, 3, 1, 0)
    out[img + "embedding/bias"] = _np(sd[VT + "embeddings.patch_embedding.bias"])
    out[img + "pos_embedding"] = _np(sd[VT + "embeddings.position_embedding.weight"])[None]
    blk = img + "Transformer/encoderblock/"
    att = blk + "MultiHeadDotProductAttention_0/"
    depth = depth_of(VT + "encoder.")
    v = lambda i, name: _np(sd[f"{VT}encoder.layers.{i}.{name}"])  # noqa: E731
    W = v(0, "self_attn.q_proj.weight").shape[1]
    h = siglip_heads
    for j, name in ((0, "layer_norm1"), (1, "layer_norm2")):
        out[f"{blk}LayerNorm_{j}/scale"] = np.stack([v(i, name + ".weight") for i in range(depth)])
        out[f"{blk}LayerNorm_{j}/bias"] = np.stack([v(i, name + ".bias") for i in range(depth)])
    for jn, tn in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj")):
        out[f"{att}{jn}/kernel"] = np.stack([v(i, f"self_attn.{tn}.weight").T.reshape(W, h, -1) for i in range(depth)])
        out[f"{att}{jn}/bias"] = np.stack([v(i, f"self_attn.{tn}.bias").reshape(h, -1) for i in range(depth)])
    out[att + "out/kernel"] = np.stack([v(i, "self_attn.out_proj.weight").T.reshape(h, -1, W) for i in range(depth)])
    out[att + "out/bias"] = np.stack([v(i, "self_attn.out_proj.bias") for i in range(depth)])
    for j, name in ((0, "fc1"), (1, "fc2")):
        out[f"{blk}MlpBlock_0/Dense_{j}/kernel"] = np.stack([v(i, f"mlp.{name}.weight").T for i in range(depth)])
        out[f"{blk}MlpBlock_0/Dense_{j}/bias"] = np.stack([v(i, f"mlp.{name}.bias") for i in range(depth)])
    out[img + "Transformer/encoder_norm/scale"] = _np(sd[VT + "post_layernorm.weight"])
    out[img + "Transformer/encoder_norm/bias"] = _np(sd[VT + "post_layernorm.bias"])
    out[img + "head/kernel"] = _np(sd[PROJ + "weight"]).T
    out[img + "head/bias"] = _np(sd[PROJ + "bias"])
    for name in HEADS:
        if name + ".weight" in sd:
            out[f"{name}/kernel"] = _np(sd[name + ".weight"]).T
            out[f"{name}/bias"] = _np(sd[name + ".bias"])
    return out


# ------------------------------------------------------------------------------------------ Orbax directory I/O (optional)
def restore_params(params_path, *, dtype=None) -> dict:
    """`openpi.models.model.restore_params(params_path, restore_type=np.ndarray)` (models/model.py:319-365): the `params` item of
    an Orbax PyTree checkpoint as a nested dict of numpy arrays, the trailing "value" level that `nnx.State` adds to every
    leaf of training checkpoints removed.  Needs `orbax-checkpoint` (not part of this image: the import is deferred, and
    everything downstream — `jax_to_torch`, the safetensors writer — works on the returned tree without it)."""
    import pathlib

    try:
        import orbax.checkpoint as ocp  # type: ignore
    except ImportError as e:
        raise ImportError("restore_params reads Orbax checkpoints and needs the `orbax-checkpoint` package (pip install "
                          "orbax-checkpoint); alternatively export the tree with numpy on a machine that has it and pass the "
                          "dict to kai0_amd.convert.jax_to_torch") from e  # fmt: skip
    params_path = pathlib.Path(params_path).resolve()

    def tree_map(fn, t):
        return {k: tree_map(fn, v) for k, v in t.items()} if isinstance(t, Mapping) else fn(t)

    with ocp.PyTreeCheckpointer() as ckptr:
        metadata = ckptr.metadata(params_path)
        item = {"params": metadata["params"]}
        restore_args = tree_map(lambda _: ocp.ArrayRestoreArgs(restore_type=np.ndarray, dtype=dtype), item)
        params = ckptr.restore(params_path, ocp.args.PyTreeRestore(item=item, restore_args=restore_args))["params"]
    flat = flatten_params(params)
    if flat and all(k.split("/")[-1] == "value" for k in flat):
        flat = {k.rsplit("/", 1)[0]: v for k, v in flat.items()}
    out: dict = {}
    for k, v in flat.items():
        node = out
        parts = k.split("/")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = np.asarray(v)
    return out


def convert_checkpoint(jax_checkpoint_dir, out_dir, *, precision: str = "bfloat16", config=None) -> str:
    """JAX checkpoint directory (`<dir>/params` Orbax tree, `<dir>/assets` norm stats) -> torch checkpoint directory
    (`model.safetensors` + `assets/`) that `create_trained_policy` / `Trainer.load_checkpoint` read.  The state dict goes
    through the model once (`load_state_dict(strict=True)` + `to_bfloat16_for_selected_params`), so key set, shapes and storage
    dtypes are those of the contract."""
    import pathlib
    import shutil

    from .checkpoint import save_model_safetensors
    from .config import Pi0Config
    from .model import PI0Pytorch

    src, dst = pathlib.Path(jax_checkpoint_dir), pathlib.Path(out_dir)
    tree = restore_params(src / "params")
    cfg = config or Pi0Config()
    sd = jax_to_torch(tree, fill_missing=True, vocab_size=cfg.vocab_size)
    with torch.device("meta"):
        model = PI0Pytorch(cfg)
    model.to_empty(device="cpu")
    model.load_state_dict({k: v.to(model.state_dict()[k].dtype) if precision == "bfloat16" else v for k, v in sd.items()},
                          strict=True)  # fmt: skip
    model.paligemma_with_expert.to_bfloat16_for_selected_params(precision)
    dst.mkdir(parents=True, exist_ok=True)
    save_model_safetensors(model, str(dst / "model.safetensors"))
    if (src / "assets").exists():
        shutil.copytree(src / "assets", dst / "assets", dirs_exist_ok=True)
    return str(dst)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description="Convert an openpi JAX checkpoint (params + assets) into a torch checkpoint directory")
    ap.add_argument("jax_checkpoint_dir")
    ap.add_argument("out_dir")
    ap.add_argument("--precision", default="bfloat16", choices=["bfloat16", "float32"])
    a = ap.parse_args()
    print(convert_checkpoint(a.jax_checkpoint_dir, a.out_dir, precision=a.precision))
