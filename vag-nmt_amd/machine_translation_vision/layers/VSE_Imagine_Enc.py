"""Visual grounding module, drop-in for layers/VSE_Imagine_Enc.py of the reference."""
import torch
import torch.nn as nn

from vagnmt_hip import ops


class ImagineAttn(nn.Module):
    """Image-conditioned attention over encoder states (layers/VSE_Imagine_Enc.py:10-79)."""

    def __init__(self, method, context_size, shared_embedding_size):
        super(ImagineAttn, self).__init__()
        if method not in ("dot", "mlp"):
            raise ValueError("imagine attention method must be 'dot' or 'mlp'")
        self.method = method
        self.embedding_size = shared_embedding_size
        self.context_size = context_size
        self.mid_dim = self.context_size
        self.ctx2ctx = nn.Linear(self.context_size, self.context_size, bias=False)
        self.emb2ctx = nn.Linear(self.embedding_size, self.context_size, bias=False)
        if self.method == "mlp":
            self.mlp = nn.Linear(self.mid_dim, 1, bias=False)

    def attend_bm(self, image_vec, enc_bm, mask_bm):
        """enc_bm (B,T,C), mask_bm (B,T) -> alpha (B,T), context (B,C)."""
        return ops.ImagineAttnCtx.apply(image_vec, enc_bm, mask_bm, self.ctx2ctx.weight, self.emb2ctx.weight,
                                        self.mlp.weight if self.method == "mlp" else None,
                                        1 if self.method == "mlp" else 0)

    def forward(self, image_vec, decoder_hidden, ctx_mask=None):
        """image_vec (B,E); decoder_hidden (T,B,C); ctx_mask (T,B) -> attention weights (B,1,T)."""
        enc = decoder_hidden.transpose(0, 1).contiguous()
        if ctx_mask is None:
            mask = torch.ones(enc.shape[0], enc.shape[1], dtype=torch.float32, device=enc.device)
        else:
            mask = ctx_mask.t().contiguous().float()
        alpha, _ = self.attend_bm(image_vec, enc, mask)
        return alpha.unsqueeze(1)


class VSE_Imagine_Enc(nn.Module):
    """im_emb = l2norm(tanh(W_im im + b)); alpha = imagine_attn; ctx = alpha . enc;
    txt_emb = l2norm(tanh(W_txt ctx + b)); loss = criterion(im_emb, txt_emb).
    Same constructor / forward / get_emb_vec / get_imagine_weights as layers/VSE_Imagine_Enc.py:82-185."""

    def __init__(self, attn_type, im_size, hidden_size, shared_embedding_size, dropout_im_emb=0.0, dropout_txt_emb=0.0,
                 activation_vse=True):
        super(VSE_Imagine_Enc, self).__init__()
        self.attn_type = attn_type
        self.im_size = im_size
        self.hidden_size = hidden_size
        self.shared_embedding_size = shared_embedding_size
        # the reference overwrites both dropouts with 0.0 (VSE_Imagine_Enc.py:95-96): the branches are dead
        self.dropout_im_emb = 0.0
        self.dropout_txt_emb = 0.0
        self.activation_vse = activation_vse
        self.imagine_attn = ImagineAttn(self.attn_type, self.hidden_size, self.shared_embedding_size)
        self.im_embedding = nn.Linear(self.im_size, self.shared_embedding_size)
        self.text_embedding = nn.Linear(self.hidden_size, self.shared_embedding_size)

    def embed_bm(self, im_var, enc_bm, mask_bm):
        """-> im_emb (B,S), txt_emb (B,S), alpha (B,T), ctx (B,C) on batch-major encoder states."""
        act = 1 if self.activation_vse else 0
        im_emb = ops.ImgProjL2.apply(im_var, self.im_embedding.weight, self.im_embedding.bias, act)
        alpha, ctx = self.imagine_attn.attend_bm(im_emb, enc_bm, mask_bm)
        txt_emb = ops.ImgProjL2.apply(ctx, self.text_embedding.weight, self.text_embedding.bias, act)
        return im_emb, txt_emb, alpha, ctx

    @staticmethod
    def _bm(decoder_hiddens, context_mask):
        enc = decoder_hiddens.transpose(0, 1).contiguous()
        if context_mask is None:
            mask = torch.ones(enc.shape[0], enc.shape[1], dtype=torch.float32, device=enc.device)
        else:
            mask = context_mask.t().contiguous().float()
        return enc, mask

    def forward(self, im_var, decoder_hiddens, criterion_vse=None, context_mask=None):
        """im_var (B,D_im); decoder_hiddens (T,B,C); returns (loss_vse, context_vec (B,C))."""
        enc, mask = self._bm(decoder_hiddens, context_mask)
        return self.forward_bm(im_var, enc, mask, criterion_vse)

    def forward_bm(self, im_var, enc_bm, mask_bm, criterion_vse=None):
        loss_vse = 0
        im_emb, txt_emb, _, ctx = self.embed_bm(im_var, enc_bm, mask_bm)
        if criterion_vse is not None:
            loss_vse = criterion_vse(im_emb, txt_emb)
        return loss_vse, ctx

    def get_emb_vec(self, im_var, decoder_hiddens, ctx_mask=None):
        enc, mask = self._bm(decoder_hiddens, ctx_mask)
        im_emb, txt_emb, _, _ = self.embed_bm(im_var, enc, mask)
        return im_emb, txt_emb

    def get_imagine_weights(self, im_var, decoder_hiddens, ctx_mask=None):
        enc, mask = self._bm(decoder_hiddens, ctx_mask)
        _, _, alpha, _ = self.embed_bm(im_var, enc, mask)
        return alpha.unsqueeze(1)
