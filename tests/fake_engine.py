"""TEST-ONLY stand-in for mvlpt_amd.engine.Engine backed by the CPU oracle, so that the host-side logic around the
towers (autograd bridge, class sharding, trainer step, gradient all-reduce) can be exercised without a GPU under
gloo.  It is never importable from the product package."""
import torch

from oracle import clip_oracle as O
from mvlpt_amd.model import SyntheticTokenizer
from mvlpt_amd.weights import _randn


class OracleEngine:
    def trim(self):
        pass

    def __init__(self, sd, arch):
        self.sd, self.arch = sd, arch

    def set_vpt_dropout(self, masks):
        self._vpt_masks = masks

    def image_fwd(self, image, vpt=None, vpt_deep=None, save_for_bwd=False):
        masks, self._vpt_masks = getattr(self, "_vpt_masks", None), None      # one-shot, as in the HIP engine
        with torch.no_grad():
            feat, self._ictx = O.image_encoder_fwd(self.sd, image, None if vpt is None else vpt.detach().reshape(1, -1, vpt.shape[-1]),
                                                   None if vpt_deep is None else vpt_deep.detach(),
                                                   heads=self.arch.vision_heads, need_bwd=save_for_bwd,
                                                   vpt_masks=masks)
        return feat

    def image_bwd(self, dfeat):
        with torch.no_grad():
            dv, dd = O.image_encoder_bwd(self.sd, dfeat, self._ictx)
        return (None if dv is None else dv[0]), dd

    def text_fwd(self, prefix, suffix, ctx, layout, eot, save_for_bwd=False):
        with torch.no_grad():
            c = None if ctx is None else ctx.detach()
            prompts = O.assemble_prompts(c, prefix, suffix, layout)
            feat, self._tctx = O.text_encoder_fwd(self.sd, prompts, eot.long(), heads=self.arch.transformer_heads,
                                                  need_bwd=save_for_bwd)
            self._layout, self._ctx_shape = layout, (None if c is None else tuple(c.shape))
        return feat

    def text_bwd(self, dfeat):
        with torch.no_grad():
            return O.scatter_prompt_grad(O.text_encoder_bwd(self.sd, dfeat, self._tctx), self._layout, self._ctx_shape)

    def logits_fwd(self, img, txt, scale, lo=None, hi=None):
        mask = None
        if lo is not None:
            idx = torch.arange(txt.shape[0]).unsqueeze(0)
            mask = ((idx >= lo.unsqueeze(-1)) & (idx < hi.unsqueeze(-1))).float()
        logits, self._lctx = O.logits_fwd(img, txt, scale, mask)
        return logits

    def logits_bwd(self, dlogits, need_img=True, need_txt=True):
        dimg, dtxt = O.logits_bwd(dlogits, self._lctx)
        return (dimg if need_img else None), (dtxt if need_txt else None)

    def cross_entropy(self, logits, label, need_grad=True):
        loss, dl = O.cross_entropy_fwd_bwd(logits, label)
        tgt = label if label.dtype == torch.int64 else label.argmax(-1)
        nc = (logits.argmax(-1) == tgt).float().sum().reshape(1)
        return loss.reshape(1), (dl if need_grad else None), nc


class OracleFrozenCLIP:
    """Duck-typed FrozenCLIP (mvlpt_amd/model.py) on the CPU oracle."""

    def __init__(self, sd, arch, token_seed=0):
        self.arch, self.engine = arch, OracleEngine(sd, arch)
        self.device = torch.device("cpu")
        self.context_length = arch.context_length
        self.logit_scale = sd["logit_scale"].float()
        self.tokenizer = SyntheticTokenizer()
        self.dtype = torch.float32
        self._emb = _randn("token_embedding.weight", token_seed, (arch.vocab_size, arch.transformer_width), 0.02)

    def token_embedding(self, ids):
        return self._emb[ids]
