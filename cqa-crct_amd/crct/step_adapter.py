"""Step adapter: the batch-dict -> model -> loss function the reference's train / eval loops call.

Mirror of ``forward()`` and ``sequence_mask()`` of CRCT/backbone/encoder_decorator.py:57-158 -- same
arguments, same return tuples (train: 7-tuple, evaluation: 6-tuple with ``loss=None``), same loss
combination ``nsp_loss_coeff * nsp + reg_loss_coeff * mean_B(reg_loss)``.  Host tensors of the batch
are moved to ``params['device']`` with non-blocking copies; the key-length mask is built on the
host side of the boundary exactly as the reference does (it depends only on integer indices).
"""
import torch

from .model import SequenceMask


def sequence_mask(sequence_length, max_len=None):
    """True for positions < length (encoder_decorator.py:57-70)."""
    if max_len is None:
        max_len = int(sequence_length.max())
    rng = torch.arange(0, max_len, device=sequence_length.device).long()
    return rng.unsqueeze(0) < sequence_length.unsqueeze(1)


def forward(dialog_encoder, batch, params, output_nsp_scores=False, output_lm_scores=False, evaluation=False,
            sample_ids=None):
    def pick(key):
        # sample_ids=None selects every row (encoder_decorator.py:76 indexes with arange(B)): the identity gather is
        # skipped -- on device-resident batches it costs one pageable H2D copy of the index (a host sync) per key
        return batch[key] if sample_ids is None else batch[key][sample_ids]

    tokens, txt_loc, segments = pick("tokens"), pick("loc"), pick("segments")
    sep_indices, mask, hist_len = pick("sep_indices"), pick("mask"), pick("hist_len")
    features, image_loc, image_mask = pick("image_feat"), pick("image_loc"), pick("image_mask")
    R = pick("R")
    if "areas" in batch:
        raise NotImplementedError("'areas' is a figure_qa / dvqa input (encoder_decorator.py:93-96); PlotQA path only")
    next_sentence_labels = image_label = None
    if not evaluation:
        next_sentence_labels = pick("next_sentence_labels")
        image_label = pick("image_label")
        regression_target = [R, "L1_smooth"]          # :104
    else:
        regression_target = [R, "L1"]                 # :106
    image_target = pick("image_target")

    # text key mask = arange(T) < sep_indices[hist_len] + 1   (:118-120).  The native model takes the DESCRIPTION of that
    # mask and builds it inside its first launch; any other model gets the materialised tensor, as in the reference.
    core = getattr(dialog_encoder, "module", dialog_encoder)
    native = hasattr(getattr(core, "bert_pretrained", None), "flat_params")
    attention_mask = SequenceMask(sep_indices, hist_len, tokens.shape[1])
    if not native:
        attention_mask = attention_mask.materialize()
    sep_len = hist_len + 1

    lm_loss, img_loss, nsp_loss, nsp_scores, regression, legend_loss = dialog_encoder(
        tokens, txt_loc, features, image_loc, sep_indices=sep_indices, sep_len=sep_len, token_type_ids=segments,
        masked_lm_labels=mask, attention_mask=attention_mask, next_sentence_label=next_sentence_labels,
        output_nsp_scores=output_nsp_scores, output_lm_scores=output_lm_scores, image_attention_mask=image_mask,
        image_label=image_label, image_target=image_target, gt_reg=regression_target, areas=None)

    loss = None
    if not evaluation:
        fused = getattr(core.bert_pretrained, "last_loss", None) if native else None
        if fused is not None and float(params["nsp_loss_coeff"]) == core.bert_pretrained.loss_coeffs[0] and \
                float(params["reg_loss_coeff"]) == core.bert_pretrained.loss_coeffs[1]:
            loss = fused          # = nsp_loss_coeff * nsp_loss + reg_loss_coeff * mean_B(reg_loss), from the head kernel (:144-153)
        else:
            reg_loss = regression[1].mean()
            loss = (params["nsp_loss_coeff"] * nsp_loss) + (params["reg_loss_coeff"] * reg_loss)     # :145
            loss = loss.sum()
    if evaluation:
        return loss, lm_loss, nsp_loss, img_loss, nsp_scores, regression
    return loss, lm_loss, nsp_loss, img_loss, nsp_scores, regression, legend_loss
