"""Synthetic PlotQA-shaped batches and deterministic weights.

``make_batch`` produces the batch dict that the reference's DataLoader hands to
``encoder_decorator.forward`` (schema: SURVEY.md 8b "Batch schema"; sources
``CRCT/fig_dataloader.py:570-688`` and ``CRCT/utils.py:105-225``) from a seeded CPU generator, with the
distribution SURVEY.md 8d fixes for the benchmark.

``seeded_fill_`` overwrites every tensor of a state_dict with values drawn from a generator
seeded by a hash of the tensor's key, so the same weights can be rebuilt anywhere (here on the
reference model when fixtures are generated, on the GPU box for this framework) without shipping
a 1 GB checkpoint.
"""
import hashlib

import torch

CAPTION_SEGMENTS = (4, 7, 8, 9, 10, 11)


def make_batch(B, T, V, F_v, categories=228, vocab_size=30522, seed=1234, ragged=True,
               needs_reg_p=0.5, n_types=12, lengths=None, n_vis=None):
    """One CPU batch dict. ``ragged`` gives up to 4 padded text keys / 6 padded visual keys; ``lengths`` / ``n_vis`` (one
    int per row) fix the number of real tokens / visual elements instead -- the reference pads every PlotQA sample to
    max_seq_len = 124 tokens and max_vis_features = 44 elements (CRCT/utils.py:152-160, fig_dataloader.py:365-390), so real
    batches carry long runs of padding."""
    g = torch.Generator().manual_seed(int(seed))

    def randint(lo, hi, shape):
        return torch.randint(lo, hi, shape, generator=g, dtype=torch.int64)

    lo_tok = min(1000, vocab_size // 2)
    tokens = randint(lo_tok, vocab_size, (B, T))
    segments = torch.zeros(B, T, dtype=torch.int64)
    loc = torch.zeros(B, T, 4, dtype=torch.float32)
    sep_indices = torch.zeros(B, 50, dtype=torch.int64)
    if lengths is not None:
        lengths = torch.tensor([int(v) for v in lengths], dtype=torch.int64)
        assert lengths.shape == (B,) and int(lengths.min()) >= 4 and int(lengths.max()) <= T
    else:
        lengths = randint(max(T - 4, 4), T + 1, (B,)) if ragged else torch.full((B,), T, dtype=torch.int64)
    cap_choices = torch.tensor([s for s in CAPTION_SEGMENTS if s < n_types] or [2], dtype=torch.int64)
    for b in range(B):
        L = int(lengths[b])
        tokens[b, 0] = 101 % vocab_size
        tokens[b, L - 1] = 102 % vocab_size
        tokens[b, L:] = 0
        n_ans = 2
        n_q = min(max(5, (L - 1 - n_ans) // 2), max(L - 1 - n_ans, 0))
        n_cap = L - 1 - n_ans - n_q
        pos = 1
        if n_cap > 0:
            # caption prefix: runs of one OCR element type each, with a box
            while pos < 1 + n_cap:
                run = min(int(randint(1, 4, (1,))), 1 + n_cap - pos)
                seg = int(cap_choices[int(randint(0, len(cap_choices), (1,)))])
                segments[b, pos:pos + run] = seg
                loc[b, pos:pos + run] = torch.rand(4, generator=g)
                pos += run
        segments[b, pos:pos + n_q] = -1
        pos += n_q
        segments[b, pos:L] = 1
        sep_indices[b, 0] = L - 1
    hist_len = torch.zeros(B, 1, dtype=torch.int64)
    mask = torch.full((B, T), -1, dtype=torch.int64)

    image_feat = torch.randn(B, V, F_v, generator=g, dtype=torch.float32)
    image_loc = torch.rand(B, V, 4, generator=g) * 1.2 - 0.1
    image_loc[:, 0] = 0
    image_target = randint(min(8, categories - 1), categories, (B, V))
    image_target[:, 0] = categories
    if n_vis is not None:
        n_vis = torch.tensor([int(v) for v in n_vis], dtype=torch.int64)
        assert n_vis.shape == (B,) and int(n_vis.min()) >= 1 and int(n_vis.max()) <= V
    else:
        n_vis = randint(max(V - 6, 2), V + 1, (B,)) if ragged else torch.full((B,), V, dtype=torch.int64)
    image_mask = (torch.arange(V)[None, :] < n_vis[:, None]).to(torch.int64)
    image_label = torch.full((B, V), -1, dtype=torch.int64)

    nsl = (torch.rand(B, 1, generator=g) < 0.5).to(torch.int64)
    needs = torch.rand(B, 1, generator=g) < needs_reg_p
    gt = torch.rand(B, generator=g) * 100.0
    R = torch.stack([gt, needs.view(-1).float(), torch.full((B,), 0.01), torch.full((B,), 100.0)], dim=1)
    return dict(tokens=tokens, segments=segments, sep_indices=sep_indices, mask=mask, loc=loc,
                hist_len=hist_len, next_sentence_labels=nsl, R=R, needs_reg=needs,
                image_feat=image_feat, image_loc=image_loc, image_mask=image_mask,
                image_target=image_target, image_label=image_label)


def _key_seed(key, base_seed):
    h = hashlib.sha256(("%d:%s" % (base_seed, key)).encode()).digest()
    return int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF


def seeded_tensor(key, shape, base_seed=0, std=0.02):
    """Deterministic fp32 values for the parameter named ``key`` (name without any wrapper prefix).

    LayerNorm weights are 1 + N(0, std); everything else (weights *and* biases) N(0, std), so
    that parity checks are sensitive to every bias / affine term.
    """
    g = torch.Generator().manual_seed(_key_seed(key, base_seed))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std
    if "LayerNorm" in key and key.endswith("weight"):
        t += 1.0
    return t


def seeded_fill_(state_dict, base_seed=0, std=0.02, strip_prefix="bert_pretrained."):
    """In-place name-keyed fill of every floating tensor in ``state_dict``."""
    with torch.no_grad():
        for k, v in state_dict.items():
            if not torch.is_floating_point(v):
                continue
            name = k[len(strip_prefix):] if strip_prefix and k.startswith(strip_prefix) else k
            if name == "cls.predictions.decoder.weight":  # tied to the word embeddings (vilbert.py:1029)
                name = "bert.embeddings.word_embeddings.weight"
            v.copy_(seeded_tensor(name, v.shape, base_seed, std).to(v.dtype))
    return state_dict
