"""Start from a BERT-base checkpoint, as the reference's constructor does.

Reference: ``VisualDialogEncoder.__init__`` (CRCT/backbone/encoder_decorator.py:16) builds its model with
``BertForMultiModalPreTraining.from_pretrained('bert-base-uncased', config, params=params)`` --
CRCT/backbone/vilbert.py:1154-1285: resolve the archive (a name of the download map, a directory holding
``pytorch_model.bin``, a ``*.bin`` file or a ``.tar.gz`` archive, :1154-1200), ``torch.load`` it on the CPU (:1209-1215), rename
the TensorFlow-era LayerNorm names ``gamma`` / ``beta`` to ``weight`` / ``bias`` (:1219-1231), choose the key prefix (``bert.`` is
prepended when the model has no ``bert`` attribute but the checkpoint's keys start with it, :1259-1263) and copy, module by module
through ``_load_from_state_dict`` (:1243-1264), every tensor whose name AND shape match; names of the model the checkpoint lacks are
"missing" (they keep their initialisation), names of the checkpoint the model lacks are "unexpected", and a shape mismatch is a
``RuntimeError`` (:1277-1283).

There is no network here (and none on a training node should be needed): the archive is a LOCAL path.  What BERT-base supplies to the
CRCT model (config/vilbert.json): the word and position embeddings and their LayerNorm, all twelve text layers
(``bert.encoder.layer.N.*``: same names and shapes as BERT's), and the LM head (``cls.predictions.*``, tied decoder included) -- 202 of
the 561 state_dict entries; the visual stream, the connection layers, both poolers, the location / type embeddings and the regressor
have no counterpart and keep ``init_bert_weights``.

``plan_load`` is the pure-Python part (CPU, no model object needed); ``load_pretrained`` applies a plan to a ``CrctModel``.
"""
import collections
import os
import tarfile
import tempfile

import torch

WEIGHTS_NAME = "pytorch_model.bin"          # vilbert.py:1151


def resolve_archive(path):
    """vilbert.py:1183-1200 for local paths: a directory -> ``<dir>/pytorch_model.bin``; ``*.bin`` -> that file; anything else is
    taken as a ``.tar.gz`` archive holding ``pytorch_model.bin`` (extracted into a temporary directory that the caller removes).
    Returns (weights_path, tempdir or None)."""
    path = os.fspath(path)
    if os.path.isdir(path):
        return os.path.join(path, WEIGHTS_NAME), None
    if not os.path.exists(path):
        # the reference logs an error and returns None here (:1166-1176), which fails one line later on ``None.train()``; say it directly
        raise FileNotFoundError("BERT checkpoint '%s' not found (a directory holding %s, a .bin file or a .tar.gz archive; the download map of "
                                "vilbert.py:60-68 is not available offline)" % (path, WEIGHTS_NAME))
    if path.endswith("bin"):
        return path, None
    tmp = tempfile.mkdtemp()
    with tarfile.open(path, "r:gz") as archive:
        archive.extractall(tmp)
    return os.path.join(tmp, WEIGHTS_NAME), tmp


def read_state_dict(source):
    """A state dict from a path (see ``resolve_archive``), an ``nn.Module`` or a mapping; always a fresh ``OrderedDict`` of CPU tensors."""
    if isinstance(source, (str, os.PathLike)):
        weights, tmp = resolve_archive(source)
        try:
            sd = torch.load(weights, map_location="cpu", weights_only=True)
        finally:
            if tmp:
                import shutil
                shutil.rmtree(tmp, ignore_errors=True)
    else:
        sd = source
    if hasattr(sd, "state_dict") and callable(sd.state_dict):          # :1214-1215
        sd = sd.state_dict()
    return collections.OrderedDict((k, v) for k, v in sd.items())


def rename_legacy_keys(sd):
    """``gamma`` -> ``weight``, ``beta`` -> ``bias`` anywhere in a key (vilbert.py:1219-1231; both replacements are applied the way the
    reference applies them: ``beta`` wins when a key holds both words)."""
    out = collections.OrderedDict()
    for key, v in sd.items():
        new_key = None
        if "gamma" in key:
            new_key = key.replace("gamma", "weight")
        if "beta" in key:
            new_key = key.replace("beta", "bias")
        out[new_key or key] = v
    return out


def plan_load(model_shapes, checkpoint, model_has_bert=True):
    """What ``from_pretrained`` would do with ``checkpoint`` for a model whose state_dict has ``model_shapes`` ({key: shape}, keys
    without any wrapper prefix).  Returns ``(copies, missing, unexpected, errors)``: ``copies`` = [(model key, checkpoint tensor)]."""
    sd = rename_legacy_keys(checkpoint)
    prefix = ""
    if not model_has_bert and any(k.startswith("bert.") for k in sd):       # :1259-1263
        prefix = "bert."
    copies, missing, errors, used = [], [], [], set()
    for key, shape in model_shapes.items():
        src = prefix + key
        if src not in sd:
            missing.append(key)
            continue
        used.add(src)
        t = sd[src]
        if tuple(t.shape) != tuple(shape):
            # torch.nn.Module._load_from_state_dict's message
            errors.append("size mismatch for %s: copying a param with shape %s from checkpoint, the shape in current model is %s."
                          % (key, tuple(t.shape), tuple(shape)))
            continue
        copies.append((key, t))
    unexpected = [k for k in sd if k not in used and k.startswith(prefix)]
    return copies, missing, unexpected, errors


@torch.no_grad()
def load_pretrained(core, source, verbose=False):
    """Copy a BERT(-base) checkpoint into ``core`` (a ``CrctModel``) with the reference's rules; returns (missing, unexpected)."""
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in core.state_dict().items())
    copies, missing, unexpected, errors = plan_load(shapes, read_state_dict(source), model_has_bert=any(k.startswith("bert.") for k in shapes))
    if errors:
        raise RuntimeError("Error(s) in loading state_dict for %s:\n\t%s" % (type(core).__name__, "\n\t".join(errors)))      # :1277-1283
    target = core.state_dict()
    for key, t in copies:
        target[key].copy_(t.to(dtype=target[key].dtype))
    core._invalidate_shadow()
    if verbose:
        print("Weights of %s not initialized from pretrained model: %d tensors; weights from pretrained model not used: %d tensors; "
              "%d tensors loaded" % (type(core).__name__, len(missing), len(unexpected), len(copies)))
    return missing, unexpected
