"""BASELINE.json's configuration at FULL DEPTH and full width (18 joint Gemma-2B / 300M-expert layers, 27 SigLIP layers, three
224^2 cameras, 200 prompt tokens, 50 x 32 actions) for the HIP model AND the CPU oracle with identical weights.

The 2.7 B parameters are drawn ON THE GPU (seconds) following the rule of `oracle.pi0_oracle.synthetic_weights_` (SURVEY.md
§8d: N(0, 0.02) matrices / embeddings / biases, plain-RMSNorm weights 0, LayerNorm weights 1 + N(0, 0.02), adaRMS dense
non-trivial) and copied to the host once; the oracle is allocated on the meta device and filled from that copy, so no CPU
random stream over billions of elements and no nn.init pass is paid.  vocab 2048: the synthetic prompts only draw ids below
2048, the embedding table carries no FLOPs, and the dead expert lm_head would only cost host memory."""

import torch


def synthetic_weights_device_(model, seed: int = 0) -> None:
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        seen = set()
        for name, p in model.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            plain_norm = "vision" not in name and "dense" not in name and (
                (name.endswith("layernorm.weight") and p.dim() == 1) or name.endswith("norm.weight"))
            if plain_norm:
                p.zero_()
                continue
            r = torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * 0.02
            if "vision" in name and ("layer_norm" in name or "post_layernorm" in name) and name.endswith("weight"):
                r += 1.0
            p.copy_(r.to(p.dtype))


def build_hip(device, seed: int = 0):
    from kai0_amd.config import Pi0Config
    from kai0_amd.model import PI0Pytorch

    with torch.device(device):
        model = PI0Pytorch(Pi0Config(vocab_size=2048))
    synthetic_weights_device_(model, seed)
    model.train_augmentation = False
    return model


def host_state(model) -> dict:
    """state dict on the host in the stored dtypes (bf16 matrices, f32 norms / heads)"""
    return {n: v.detach().to("cpu") for n, v in model.state_dict().items()}


def oracle_from_state(state: dict, dtype: str):
    """The full-depth oracle in `dtype` ("bfloat16": the reference's mixed choreography; "float32": everything f32, the same
    bf16-representable weight values upcast) holding `state`'s values."""
    from oracle import pi0_oracle as O

    cfg = O.OracleConfig(dtype=dtype, vocab_size=2048)
    with torch.device("meta"):
        model = O.OraclePI0(cfg)
    model.to_empty(device="cpu")
    with torch.no_grad():
        model.load_state_dict(state, strict=True)  # copies into the oracle's own dtypes (f32 oracle: bf16 values upcast)
        for mod in model.modules():  # non-persistent buffers are not materialised by to_empty
            if isinstance(getattr(mod, "inv_freq", None), torch.Tensor):
                # `.to(bfloat16)` of the whole module also rounds this buffer (gemma_pytorch.py:64-65); the f32 oracle is the SAME
                # function evaluated in higher precision, so it keeps the rounded values (as tests/fullwidth.py's upcast copy does)
                inv = O.rope_inv_freq(mod.inv_freq.numel() * 2).to(torch.bfloat16)
                mod.inv_freq = inv if dtype == "bfloat16" else inv.float()
            if isinstance(mod, O.SiglipVisionEmbeddings):
                mod.position_ids = torch.arange(mod.num_patches).expand((1, -1))
    return model, cfg
