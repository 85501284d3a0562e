"""Architecture of the Transformer recogniser (app/ml/models/text_recognizer.py:39-69 loads
``microsoft/trocr-base-printed`` through ``VisionEncoderDecoderModel.from_pretrained``): a ViT encoder (HF ``ViTModel``) and
the TrOCR decoder (HF ``TrOCRForCausalLM``), greedy ``generate(max_length=50)``.

Neither the checkpoint nor its config.json can be fetched here, so the numbers below restate the published
trocr-base-printed configuration (ViT-base/16 at 384x384 without q/k/v biases; 12-layer post-LN decoder, d_model 1024,
16 heads, ffn 4096, 50265-token RoBERTa vocabulary, learned positions with offset 2, layernorm_embedding, no embedding
scale, cross-attention straight onto the 768-wide encoder states).  The engine is parametric in them; ``TINY`` is a
reduced instance for fast tests.  State-dict keys are the ones transformers==4.36.0 (the reference's pin) writes:

  encoder.embeddings.{cls_token,position_embeddings,patch_embeddings.projection.{weight,bias}}
  encoder.encoder.layer.N.{layernorm_before,layernorm_after}.{weight,bias}
  encoder.encoder.layer.N.attention.attention.{query,key,value}.weight[,bias]
  encoder.encoder.layer.N.attention.output.dense.{weight,bias}
  encoder.encoder.layer.N.{intermediate,output}.dense.{weight,bias}
  encoder.layernorm.{weight,bias}      (encoder.pooler.* is accepted and ignored: generate() never reads it)
  decoder.model.decoder.{embed_tokens,embed_positions}.weight, decoder.model.decoder.layernorm_embedding.{weight,bias}
  decoder.model.decoder.layers.N.{self_attn,encoder_attn}.{q,k,v,out}_proj.{weight,bias}
  decoder.model.decoder.layers.N.{self_attn_layer_norm,encoder_attn_layer_norm,final_layer_norm}.{weight,bias}
  decoder.model.decoder.layers.N.{fc1,fc2}.{weight,bias}
  decoder.output_projection.weight     (absent = tied to embed_tokens)
"""
from dataclasses import asdict, dataclass


@dataclass(frozen=True)
class TrOCRSpec:
    image_size: int = 384
    patch_size: int = 16
    enc_hidden: int = 768
    enc_layers: int = 12
    enc_heads: int = 12
    enc_ffn: int = 3072
    enc_qkv_bias: bool = False
    enc_ln_eps: float = 1e-12
    dec_hidden: int = 1024
    dec_layers: int = 12
    dec_heads: int = 16
    dec_ffn: int = 4096
    vocab_size: int = 50265
    max_positions: int = 512
    dec_ln_eps: float = 1e-5
    decoder_start_token_id: int = 2
    eos_token_id: int = 2
    pad_token_id: int = 1
    max_length: int = 50          # text_recognizer.py:58: generate(pixel_values, max_length=50)

    @property
    def enc_tokens(self):
        return (self.image_size // self.patch_size) ** 2 + 1

    def as_dict(self):
        return asdict(self)


BASE_PRINTED = TrOCRSpec()
# reduced instance (same code paths: head_dim 64, token count not a multiple of 16, vocabulary not a multiple of 64)
TINY = TrOCRSpec(image_size=96, enc_hidden=128, enc_layers=2, enc_heads=2, enc_ffn=256, dec_hidden=192, dec_layers=2, dec_heads=3,
                 dec_ffn=384, vocab_size=1000, max_positions=64)


def hf5_key(key):
    """transformers 5.x renamed the ViT encoder's parameters; map a 4.36 checkpoint key to the 5.x module path (the
    golden generator loads this build's 4.36-keyed state dicts into the locally installed 5.x classes)."""
    k = key.replace("encoder.encoder.layer.", "encoder.layers.")
    for old, new in ((".attention.attention.query.", ".attention.q_proj."), (".attention.attention.key.", ".attention.k_proj."),
                     (".attention.attention.value.", ".attention.v_proj."), (".attention.output.dense.", ".attention.o_proj."),
                     (".intermediate.dense.", ".mlp.fc1."), (".output.dense.", ".mlp.fc2.")):
        if k.startswith("encoder.layers."):
            k = k.replace(old, new)
    return k


def hf4_key(key):
    """Inverse of hf5_key (accept a state dict saved by transformers 5.x)."""
    if not key.startswith("encoder.layers."):
        return key
    k = key.replace("encoder.layers.", "encoder.encoder.layer.")
    for new, old in ((".attention.attention.query.", ".attention.q_proj."), (".attention.attention.key.", ".attention.k_proj."),
                     (".attention.attention.value.", ".attention.v_proj."), (".attention.output.dense.", ".attention.o_proj."),
                     (".intermediate.dense.", ".mlp.fc1."), (".output.dense.", ".mlp.fc2.")):
        k = k.replace(old, new)
    return k
