"""fp32 torch-CPU restatement of the Transformer recogniser (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The reference (app/ml/models/text_recognizer.py:39-69) delegates every operation to third-party code that is not under
/root/reference: ``transformers==4.36.0`` (requirements.txt:13) -- ``TrOCRProcessor`` (a ``ViTImageProcessor``: PIL
bilinear resize to 384x384, /255, normalise with mean = std = 0.5) and ``VisionEncoderDecoderModel`` (``ViTModel``
encoder, ``TrOCRForCausalLM`` decoder) driven by ``generate(pixel_values, max_length=50)`` with the checkpoint's default
greedy search.  The algorithm restated here is the published one of those classes:

  preprocess      text_recognizer.py:48-55  cvtColor(BGR2RGB) -> PIL -> processor(...)  (Pillow's resample is pinned
                                            bit-exact by oracle/csrc/oracle.c:orc_pil_resize_bilinear)
  encode          ViTModel.forward          patch conv 16x16/s16 + [CLS] + learned positions; 12 pre-LN blocks
                                            (LN -> MHA -> +res; LN -> fc1 -> exact-erf GELU -> fc2 -> +res); final LN
  decode_step     TrOCRDecoder.forward      embed_tokens + learned positions (offset 2) -> layernorm_embedding; 12 post-LN
                                            blocks (self-attn with KV cache, cross-attn on the encoder states, fc1-GELU-fc2,
                                            q scaled by head_dim**-0.5 after its bias); output_projection (no bias)
  generate        GenerationMixin greedy    starts from decoder_start_token_id (= 2), arg-max (lowest index on ties, as
                                            torch.argmax), stops at eos (= 2) or at max_length tokens in total; rows that
                                            finished earlier are padded with pad_token_id (= 1)

PINNED against the locally installed transformers classes (5.x; the modelling maths of these two models is unchanged since
4.36, only parameter names moved) on seeded weights: tests/golden/make_golden_trocr.py -> tests/golden/trocr_*.npz,
checked by tests/test_oracle_trocr.py.  PARITY UNPINNED against the real microsoft/trocr-base-printed checkpoint and its
tokenizer (neither can be fetched): token ids are compared, not strings.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cstages


def preprocess(crop_bgr, spec):
    """uint8 HxWx3 BGR crop -> float32 [3,S,S] pixel_values."""
    rgb = np.ascontiguousarray(crop_bgr[..., ::-1])
    small = cstages.pil_resize_bilinear(rgb, spec.image_size, spec.image_size)
    x = torch.from_numpy(small).permute(2, 0, 1).float() / 255.0
    return (x - 0.5) / 0.5


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _heads(x, h):
    b, t, c = x.shape
    return x.view(b, t, h, c // h).transpose(1, 2)


@torch.no_grad()
def encode(pixel_values, sd, spec, return_taps=False):
    """[B,3,S,S] -> encoder last_hidden_state [B, tokens, enc_hidden] (after the final LayerNorm)."""
    e = "encoder.embeddings."
    x = F.conv2d(pixel_values, sd[e + "patch_embeddings.projection.weight"], sd[e + "patch_embeddings.projection.bias"],
                 stride=spec.patch_size)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[e + "cls_token"].expand(x.shape[0], -1, -1), x], dim=1) + sd[e + "position_embeddings"]
    taps = [x]
    for i in range(spec.enc_layers):
        p = f"encoder.encoder.layer.{i}."
        y = _ln(x, sd, p + "layernorm_before", spec.enc_ln_eps)
        q, k, v = (_heads(_lin(y, sd, p + "attention.attention." + n), spec.enc_heads) for n in ("query", "key", "value"))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), dim=-1) @ v
        a = a.transpose(1, 2).reshape(x.shape)
        x = x + _lin(a, sd, p + "attention.output.dense")
        y = _ln(x, sd, p + "layernorm_after", spec.enc_ln_eps)
        x = x + _lin(F.gelu(_lin(y, sd, p + "intermediate.dense")), sd, p + "output.dense")
        taps.append(x)
    out = _ln(x, sd, "encoder.layernorm", spec.enc_ln_eps)
    return (out, taps) if return_taps else out


class _Decoder:
    def __init__(self, enc, sd, spec):
        self.sd, self.spec, self.enc = sd, spec, enc
        self.q = "decoder.model.decoder."
        self.scale = (spec.dec_hidden // spec.dec_heads) ** -0.5
        self.cross = []
        for i in range(spec.dec_layers):
            p = self.q + f"layers.{i}.encoder_attn."
            self.cross.append((_heads(_lin(enc, sd, p + "k_proj"), spec.dec_heads), _heads(_lin(enc, sd, p + "v_proj"), spec.dec_heads)))
        self.kv = [[None, None] for _ in range(spec.dec_layers)]
        self.pos = 0

    def _attend(self, q, k, v):
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        return a.transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)

    def step(self, tokens, return_hidden=False):
        """tokens [B] (the newest token of every row) -> logits [B, vocab] of the next one."""
        sd, spec, q = self.sd, self.spec, self.q
        x = sd[q + "embed_tokens.weight"][tokens] + sd[q + "embed_positions.weight"][self.pos + 2]
        x = _ln(x, sd, q + "layernorm_embedding", spec.dec_ln_eps).unsqueeze(1)
        for i in range(spec.dec_layers):
            p = q + f"layers.{i}."
            qs = _heads(_lin(x, sd, p + "self_attn.q_proj") * self.scale, spec.dec_heads)
            k = _heads(_lin(x, sd, p + "self_attn.k_proj"), spec.dec_heads)
            v = _heads(_lin(x, sd, p + "self_attn.v_proj"), spec.dec_heads)
            if self.kv[i][0] is not None:
                k, v = torch.cat([self.kv[i][0], k], dim=2), torch.cat([self.kv[i][1], v], dim=2)
            self.kv[i] = [k, v]
            x = _ln(x + _lin(self._attend(qs, k, v), sd, p + "self_attn.out_proj"), sd, p + "self_attn_layer_norm", spec.dec_ln_eps)
            qc = _heads(_lin(x, sd, p + "encoder_attn.q_proj") * self.scale, spec.dec_heads)
            x = _ln(x + _lin(self._attend(qc, *self.cross[i]), sd, p + "encoder_attn.out_proj"), sd, p + "encoder_attn_layer_norm",
                    spec.dec_ln_eps)
            x = _ln(x + _lin(F.gelu(_lin(x, sd, p + "fc1")), sd, p + "fc2"), sd, p + "final_layer_norm", spec.dec_ln_eps)
        self.pos += 1
        w = sd.get("decoder.output_projection.weight", sd[q + "embed_tokens.weight"])
        logits = F.linear(x[:, 0], w)
        return (logits, x[:, 0]) if return_hidden else logits


@torch.no_grad()
def generate(enc, sd, spec, max_length=None, forced=None):
    """Greedy search on encoder states [B,T,C].  Returns (ids [B, L] int64 incl. the start token, padded with pad_token_id;
    logits [B, L-1, vocab] of every step that was run).  ``forced`` ([B, L] ids): teacher forcing -- feed these tokens
    instead of the arg-max (tensor-level tests compare logits on a fixed token path)."""
    max_length = max_length or spec.max_length
    b = enc.shape[0]
    dec = _Decoder(enc, sd, spec)
    ids = torch.full((b, 1), spec.decoder_start_token_id, dtype=torch.long)
    done = torch.zeros(b, dtype=torch.bool)
    all_logits = []
    while ids.shape[1] < max_length:
        logits = dec.step(ids[:, -1])
        all_logits.append(logits)
        nxt = logits.argmax(dim=-1)
        if forced is not None:
            if ids.shape[1] >= forced.shape[1]:
                break
            nxt = forced[:, ids.shape[1]]
        nxt = torch.where(done, torch.full_like(nxt, spec.pad_token_id), nxt)
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        done = done | (nxt == spec.eos_token_id)
        if forced is None and bool(done.all()):
            break
    return ids, torch.stack(all_logits, dim=1)


def recognize_ids(crops_bgr, sd, spec):
    """The reference's per-image call (text_recognizer.py:45-60): one generate per crop -> list of id lists."""
    out = []
    for c in crops_bgr:
        enc = encode(preprocess(c, spec).unsqueeze(0), sd, spec)
        out.append(generate(enc, sd, spec)[0][0].tolist())
    return out
