"""timm ViT -> SiT weight import (reference utils/utils.py:11-35, called from
tools/train.py:221-224 and tools/pretrain.py:229-232 with `timm.create_model(name, pretrained=True)`).

The reference copies the encoder of an ImageNet ViT (timm key layout `blocks.{i}.*`, final `norm`) into the
SiT state dict; patch embedding, cls token, positional embedding and the regression head weight keep their
own initialisation.  The same mapping is expressed here as one table of key templates so that the state-dict
surface of sitk.models.sit.SiT is checked against it in tests/test_host_cpu.py.  Fetching
the timm weights needs network access and is the caller's business; this function only moves tensors.
"""

# SiT key template <- timm key template ({i} = block index)
TIMM_TO_SIT = (
    ("transformer.layers.{i}.0.norm.weight", "blocks.{i}.norm1.weight"),
    ("transformer.layers.{i}.0.norm.bias", "blocks.{i}.norm1.bias"),
    ("transformer.layers.{i}.1.norm.weight", "blocks.{i}.norm2.weight"),
    ("transformer.layers.{i}.1.norm.bias", "blocks.{i}.norm2.bias"),
    ("transformer.layers.{i}.0.fn.to_qkv.weight", "blocks.{i}.attn.qkv.weight"),     # the timm qkv bias has no SiT slot
    ("transformer.layers.{i}.0.fn.to_out.0.weight", "blocks.{i}.attn.proj.weight"),
    ("transformer.layers.{i}.0.fn.to_out.0.bias", "blocks.{i}.attn.proj.bias"),
    ("transformer.layers.{i}.1.fn.net.0.weight", "blocks.{i}.mlp.fc1.weight"),
    ("transformer.layers.{i}.1.fn.net.0.bias", "blocks.{i}.mlp.fc1.bias"),
    ("transformer.layers.{i}.1.fn.net.3.weight", "blocks.{i}.mlp.fc2.weight"),
    ("transformer.layers.{i}.1.fn.net.3.bias", "blocks.{i}.mlp.fc2.bias"),
)
HEAD_NORM = (("mlp_head.0.weight", "norm.weight"), ("mlp_head.0.bias", "norm.bias"))


def load_weights_imagenet(state_dict, state_dict_imagenet, nb_layers):
    """Same call signature and result as the reference helper: returns `state_dict` with the encoder blocks
    0..nb_layers-1 and the head LayerNorm replaced by the timm tensors (shapes are checked)."""
    pairs = list(HEAD_NORM) + [(d.format(i=i), s.format(i=i)) for i in range(nb_layers) for d, s in TIMM_TO_SIT]
    for dst, src in pairs:
        if src not in state_dict_imagenet:
            raise KeyError(f"timm state dict has no '{src}' (needed for '{dst}')")
        t = state_dict_imagenet[src].data
        if dst in state_dict and tuple(state_dict[dst].shape) != tuple(t.shape):
            raise ValueError(f"shape mismatch for '{dst}': SiT {tuple(state_dict[dst].shape)} vs timm '{src}' {tuple(t.shape)}")
        state_dict[dst] = t
    return state_dict
