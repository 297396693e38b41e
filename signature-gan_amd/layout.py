"""Parameter / buffer layout of the two networks, in the reference's ``parameters()`` and
``state_dict()`` order (generator_vanilla_gan.py:124-163, discriminator_vanilla_gan.py:131-207).
The flat arenas handed to the C ABI are these tensors concatenated in this order."""
from collections import OrderedDict

G_CHAIN = {64: (256, 128, 64, 32, 32), 128: (512, 256, 128, 64, 32, 32)}
D_CHAIN = {64: (64, 128, 256, 512), 128: (64, 128, 256, 512, 512)}


def check_size(size, what="output_size"):
    if size not in (64, 128):
        raise ValueError(f"{what} must be 64 or 128, got {size}")


def generator_entries(latent_dim, size, channels=1):
    """[(state_dict key, shape, kind)] with kind in {'param', 'bn_mean', 'bn_var', 'bn_count'}."""
    check_size(size)
    chain = G_CHAIN[size]
    feat = chain[0] * 16
    e = [("fc.0.weight", (feat, latent_dim), "param"), ("fc.0.bias", (feat,), "param"),
         ("fc.1.weight", (feat,), "param"), ("fc.1.bias", (feat,), "param"),
         ("fc.1.running_mean", (feat,), "bn_mean"), ("fc.1.running_var", (feat,), "bn_var"),
         ("fc.1.num_batches_tracked", (), "bn_count")]
    for i in range(len(chain) - 1):
        p = f"upsample_blocks.{i}.block."
        e += [(p + "0.weight", (chain[i], chain[i + 1], 4, 4), "param"),
              (p + "1.weight", (chain[i + 1],), "param"), (p + "1.bias", (chain[i + 1],), "param"),
              (p + "1.running_mean", (chain[i + 1],), "bn_mean"), (p + "1.running_var", (chain[i + 1],), "bn_var"),
              (p + "1.num_batches_tracked", (), "bn_count")]
    e += [("final_conv.0.weight", (channels, chain[-1], 3, 3), "param"), ("final_conv.0.bias", (channels,), "param")]
    return e


def discriminator_entries(size, channels=1):
    check_size(size, "input_size")
    chain = (channels,) + D_CHAIN[size]
    e = []
    for i in range(len(chain) - 1):
        p = f"conv_blocks.{i}.block.0."
        e += [(p + "weight", (chain[i + 1], chain[i], 4, 4), "param"), (p + "bias", (chain[i + 1],), "param")]
    e += [("classifier.0.weight", (1, chain[-1] * 16), "param"), ("classifier.0.bias", (1,), "param")]
    return e


def numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def spans(entries, kind="param"):
    """OrderedDict key -> (offset, numel, shape) inside the flat arena of that kind."""
    out, off = OrderedDict(), 0
    for k, shape, kd in entries:
        if kd == kind:
            out[k] = (off, numel(shape), shape)
            off += numel(shape)
    return out, off
