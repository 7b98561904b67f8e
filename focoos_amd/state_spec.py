"""State-dict layouts of the RT-DETR family (``fai-detr-*``), the MaskFormer family (``fai-mf-*``; ResNet-vd backbones) and the
BiSeNetFormer family (``bisenetformer-*``; STDC backbone).

The engine keeps the reference's checkpoint key names unchanged so that a
``model_final.pth`` written by the reference loads into it and vice versa
("state_dict() with unchanged key names").  The names below are
what ``FAIDetr(config).state_dict()`` yields in the reference
(focoos/models/fai_detr/modelling.py:1273-1334, focoos/nn/backbone/resnet.py:164-250);
``tests/golden/detr_l_state_keys.json`` is a dump of the reference's own keys and
``tests/test_state_spec.py`` pins this generator against it.

kinds: conv_w, bn_w, bn_b, bn_mean, bn_var, bn_nbt, lin_w, lin_b, ln_w, ln_b, emb, buf
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

Spec = "OrderedDict[str, Tuple[Tuple[int, ...], str]]"

RESNET_BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}


def _conv_bn(spec, prefix, cin, cout, k, conv_name="conv", norm_name="norm"):
    spec[f"{prefix}.{conv_name}.weight"] = ((cout, cin, k, k), "conv_w")
    _bn(spec, f"{prefix}.{norm_name}", cout)


def _bn(spec, prefix, c):
    spec[f"{prefix}.weight"] = ((c,), "bn_w")
    spec[f"{prefix}.bias"] = ((c,), "bn_b")
    spec[f"{prefix}.running_mean"] = ((c,), "bn_mean")
    spec[f"{prefix}.running_var"] = ((c,), "bn_var")
    spec[f"{prefix}.num_batches_tracked"] = ((), "bn_nbt")


def _linear(spec, prefix, cin, cout):
    spec[f"{prefix}.weight"] = ((cout, cin), "lin_w")
    spec[f"{prefix}.bias"] = ((cout,), "lin_b")


def _ln(spec, prefix, c):
    spec[f"{prefix}.weight"] = ((c,), "ln_w")
    spec[f"{prefix}.bias"] = ((c,), "ln_b")


def _mha(spec, prefix, c):
    spec[f"{prefix}.in_proj_weight"] = ((3 * c, c), "lin_w")
    spec[f"{prefix}.in_proj_bias"] = ((3 * c,), "lin_b")
    _linear(spec, f"{prefix}.out_proj", c, c)


def resnet_vd_spec(spec, prefix: str, depth: int, in_chans: int = 3):
    """ResNet-vd (bottleneck) — focoos/nn/backbone/resnet.py:164-250."""
    if depth not in RESNET_BLOCKS:
        raise ValueError(f"unsupported resnet depth {depth} (engine covers bottleneck depths {sorted(RESNET_BLOCKS)})")
    _conv_bn(spec, f"{prefix}.conv1.conv1_1", in_chans, 32, 3)
    _conv_bn(spec, f"{prefix}.conv1.conv1_2", 32, 32, 3)
    _conv_bn(spec, f"{prefix}.conv1.conv1_3", 32, 64, 3)
    ch_in = 64
    for si, (nblk, width) in enumerate(zip(RESNET_BLOCKS[depth], [64, 128, 256, 512])):
        for bi in range(nblk):
            p = f"{prefix}.res_layers.{si}.blocks.{bi}"
            _conv_bn(spec, f"{p}.branch2a", ch_in, width, 1)
            _conv_bn(spec, f"{p}.branch2b", width, width, 3)
            _conv_bn(spec, f"{p}.branch2c", width, width * 4, 1)
            if bi == 0:
                if si == 0:  # stage 2: stride 1 -> plain 1x1 ConvNormLayer shortcut
                    _conv_bn(spec, f"{p}.short", ch_in, width * 4, 1)
                else:  # variant "d", stride 2: avgpool + 1x1 ConvNormLayer (resnet.py:89-100)
                    _conv_bn(spec, f"{p}.short.conv", ch_in, width * 4, 1)
                ch_in = width * 4
    return [256, 512, 1024, 2048]


def detr_state_spec(config: Dict) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """Ordered ``name -> (shape, kind)`` for a registry-style DETR config dict."""
    bb = config["backbone_config"]
    if bb.get("model_type", "resnet") not in ("resnet", "stdc"):
        raise ValueError("engine state spec covers resnet (fai-detr-l-*) and stdc (fai-detr-m-coco) backbones")
    nc = int(config["num_classes"])
    fd = int(config.get("pixel_decoder_feat_dim", 256))
    od = int(config.get("pixel_decoder_out_dim", 256))
    ffe = int(config.get("pixel_decoder_dim_feedforward", 1024))
    n_enc = int(config.get("pixel_decoder_num_encoder_layers", 1))
    hd = int(config.get("transformer_predictor_hidden_dim", 256))
    ffd = int(config.get("transformer_predictor_dim_feedforward", 1024))
    nl = int(config.get("transformer_predictor_dec_layers", 6))
    nh = int(config.get("transformer_predictor_nhead", 8))
    n_levels, n_points = 3, 4

    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    if bb.get("model_type", "resnet") == "stdc":
        chans = stdc_spec(spec, "pixel_decoder.backbone", int(bb.get("base", 64)), tuple(bb.get("layers", (4, 5, 3))), int(bb.get("block_num", 4)),
                          int(bb.get("in_chans", 3)))
    else:
        chans = resnet_vd_spec(spec, "pixel_decoder.backbone", int(bb.get("depth", 50)), int(bb.get("in_chans", 3)))
    # Encoder (modelling.py:195-291)
    for i, c in enumerate(chans[1:]):
        spec[f"pixel_decoder.input_proj.{i}.0.weight"] = ((fd, c, 1, 1), "conv_w")
        _bn(spec, f"pixel_decoder.input_proj.{i}.1", fd)
    for li in range(n_enc):
        p = f"pixel_decoder.encoder.0.layers.{li}"
        _mha(spec, f"{p}.self_attn", fd)
        _linear(spec, f"{p}.linear1", fd, ffe)
        _linear(spec, f"{p}.linear2", ffe, fd)
        _ln(spec, f"{p}.norm1", fd)
        _ln(spec, f"{p}.norm2", fd)

    def csp(prefix):
        _conv_bn(spec, f"{prefix}.conv1", 2 * fd, fd, 1)
        _conv_bn(spec, f"{prefix}.conv2", 2 * fd, fd, 1)
        for b in range(3):
            _conv_bn(spec, f"{prefix}.bottlenecks.{b}.conv1", fd, fd, 3)
            _conv_bn(spec, f"{prefix}.bottlenecks.{b}.conv2", fd, fd, 1)

    for i in range(2):
        _conv_bn(spec, f"pixel_decoder.lateral_convs.{i}", fd, fd, 1)
    for i in range(2):
        csp(f"pixel_decoder.fpn_blocks.{i}")
    for i in range(2):
        _conv_bn(spec, f"pixel_decoder.downsample_convs.{i}", fd, fd, 3)
    for i in range(2):
        csp(f"pixel_decoder.pan_blocks.{i}")
    spec["pixel_decoder.mask_features.weight"] = ((od, fd, 3, 3), "conv_w")
    spec["pixel_decoder.mask_features.bias"] = ((od,), "lin_b")
    # criterion buffer (modelling.py SetCriterion.empty_weight)
    spec["head.criterion.empty_weight"] = ((nc + 1,), "buf")
    # TransformerPredictor (modelling.py:1023-1104)
    for i in range(3):
        _conv_bn(spec, f"head.predictor.input_proj.{i}", od, hd, 1)
    for li in range(nl):
        p = f"head.predictor.decoder.layers.{li}"
        _mha(spec, f"{p}.self_attn", hd)
        _ln(spec, f"{p}.norm1", hd)
        _linear(spec, f"{p}.cross_attn.sampling_offsets", hd, nh * n_levels * n_points * 2)
        _linear(spec, f"{p}.cross_attn.attention_weights", hd, nh * n_levels * n_points)
        _linear(spec, f"{p}.cross_attn.value_proj", hd, hd)
        _linear(spec, f"{p}.cross_attn.output_proj", hd, hd)
        _ln(spec, f"{p}.norm2", hd)
        _linear(spec, f"{p}.linear1", hd, ffd)
        _linear(spec, f"{p}.linear2", ffd, hd)
        _ln(spec, f"{p}.norm3", hd)
    _linear(spec, "head.predictor.query_pos_head.layers.0", 4, 2 * hd)
    _linear(spec, "head.predictor.query_pos_head.layers.1", 2 * hd, hd)
    _linear(spec, "head.predictor.enc_output.0", hd, hd)
    _ln(spec, "head.predictor.enc_output.1", hd)
    _linear(spec, "head.predictor.enc_score_classifier", hd, nc)
    for j, (a, b) in enumerate([(hd, hd), (hd, hd), (hd, 4)]):
        _linear(spec, f"head.predictor.enc_bbox_classifier.layers.{j}", a, b)
    for li in range(nl):
        _linear(spec, f"head.predictor.dec_score_classifier.{li}", hd, nc)
    for li in range(nl):
        for j, (a, b) in enumerate([(hd, hd), (hd, hd), (hd, 4)]):
            _linear(spec, f"head.predictor.dec_bbox_classifier.{li}.layers.{j}", a, b)
    return spec


def mf_state_spec(config: Dict) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """Ordered ``name -> (shape, kind)`` of ``FAIMaskFormer(config).state_dict()``
    (focoos/models/fai_mf/modelling.py:633-710: TransformerFPN :201-338, MultiScaleMaskedTransformerDecoder :372-499,
    PredictionHeads :28-60); pinned against the reference's own key dump in tests/golden/mf_l_state_keys.json."""
    bb = config["backbone_config"]
    if bb.get("model_type", "resnet") not in ("resnet", "stdc"):
        raise ValueError("engine state spec covers resnet (fai-mf-l-*) and stdc (fai-mf-m-ade) backbones")
    nc = int(config["num_classes"])
    fd = int(config.get("pixel_decoder_feat_dim", 256))
    od = int(config.get("pixel_decoder_out_dim", 256))
    n_enc = int(config.get("pixel_decoder_transformer_layers", 0))
    ffe = int(config.get("pixel_decoder_transformer_dim_feedforward", 1024))
    hd = int(config.get("transformer_predictor_hidden_dim", 256))
    md = int(config.get("transformer_predictor_out_dim", 256))
    ffd = int(config.get("transformer_predictor_dim_feedforward", 1024))
    nl = int(config.get("transformer_predictor_dec_layers", 6))
    nq = int(config.get("num_queries", 100))

    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    if bb.get("model_type", "resnet") == "stdc":
        chans = stdc_spec(spec, "pixel_decoder.backbone", int(bb.get("base", 64)), tuple(bb.get("layers", (4, 5, 3))), int(bb.get("block_num", 4)),
                          int(bb.get("in_chans", 3)))
    else:
        chans = resnet_vd_spec(spec, "pixel_decoder.backbone", int(bb.get("depth", 50)), int(bb.get("in_chans", 3)))
    P = "pixel_decoder"
    if n_enc > 0:
        spec[f"{P}.input_proj.weight"] = ((fd, chans[-1], 1, 1), "conv_w")
        spec[f"{P}.input_proj.bias"] = ((fd,), "lin_b")
        for li in range(n_enc):
            p = f"{P}.transformer.encoder.layers.{li}"
            _mha(spec, f"{p}.self_attn", fd)
            _linear(spec, f"{p}.linear1", fd, ffe)
            _linear(spec, f"{p}.linear2", ffe, fd)
            _ln(spec, f"{p}.norm1", fd)
            _ln(spec, f"{p}.norm2", fd)
        _ln(spec, f"{P}.transformer.encoder.norm", fd)
    for idx, c in enumerate(chans):
        if idx < len(chans) - 1:
            spec[f"{P}.adapter_{idx + 1}.weight"] = ((fd, c, 1, 1), "conv_w")
            _bn(spec, f"{P}.adapter_{idx + 1}.norm", fd)
        spec[f"{P}.layer_{idx + 1}.weight"] = ((fd, fd if (idx < len(chans) - 1 or n_enc > 0) else c, 3, 3), "conv_w")
        _bn(spec, f"{P}.layer_{idx + 1}.norm", fd)
    spec[f"{P}.mask_features.weight"] = ((od, fd, 3, 3), "conv_w")
    spec[f"{P}.mask_features.bias"] = ((od,), "lin_b")
    spec["head.criterion.empty_weight"] = ((nc + 1,), "buf")
    H = "head.predictor"
    for li in range(nl):
        _mha(spec, f"{H}.transformer_self_attention_layers.{li}.self_attn", hd)
        _ln(spec, f"{H}.transformer_self_attention_layers.{li}.norm", hd)
    for li in range(nl):
        _mha(spec, f"{H}.transformer_cross_attention_layers.{li}.multihead_attn", hd)
        _ln(spec, f"{H}.transformer_cross_attention_layers.{li}.norm", hd)
    for li in range(nl):
        _linear(spec, f"{H}.transformer_ffn_layers.{li}.linear1", hd, ffd)
        _linear(spec, f"{H}.transformer_ffn_layers.{li}.linear2", ffd, hd)
        _ln(spec, f"{H}.transformer_ffn_layers.{li}.norm", hd)
    spec[f"{H}.query_feat.weight"] = ((nq, hd), "emb")
    spec[f"{H}.query_embed.weight"] = ((nq, hd), "emb")
    for i in range(min(3, nl)):
        spec[f"{H}.input_proj.{i}.weight"] = ((hd, od, 1, 1), "conv_w")
        spec[f"{H}.input_proj.{i}.bias"] = ((hd,), "lin_b")
    _ln(spec, f"{H}.forward_prediction_heads.decoder_norm", hd)
    _linear(spec, f"{H}.forward_prediction_heads.classifier", hd, nc + 1)
    for j, (a, b) in enumerate([(hd, hd), (hd, hd), (hd, md)]):
        _linear(spec, f"{H}.forward_prediction_heads.mask_classifier.layers.{j}", a, b)
    return spec


def stdc_spec(spec, prefix: str, base: int = 64, layers=(4, 5, 3), block_num: int = 4, in_chans: int = 3):
    """STDC backbone with CatBottleneck blocks (focoos/nn/backbone/stdc.py:108-166, 282-311); returns the res2..res5 channels."""
    if block_num != 4:
        raise ValueError("engine state spec covers STDC block_num=4 (stdc small / large)")
    _conv_bn(spec, f"{prefix}.features.0", in_chans, base // 2, 3, "conv", "bn")
    _conv_bn(spec, f"{prefix}.features.1", base // 2, base, 3, "conv", "bn")
    idx, cin = 2, base
    for i, n in enumerate(layers):
        cout = base * 2 ** (i + 2)
        for j in range(n):
            p = f"{prefix}.features.{idx}"
            _conv_bn(spec, f"{p}.conv_list.0", cin, cout // 2, 1, "conv", "bn")
            _conv_bn(spec, f"{p}.conv_list.1", cout // 2, cout // 4, 3, "conv", "bn")
            _conv_bn(spec, f"{p}.conv_list.2", cout // 4, cout // 8, 3, "conv", "bn")
            _conv_bn(spec, f"{p}.conv_list.3", cout // 8, cout // 8, 3, "conv", "bn")
            if j == 0:  # stride-2 block: depthwise 3x3 stride-2 conv + BN on the first branch
                spec[f"{p}.avd_layer.0.weight"] = ((cout // 2, 1, 3, 3), "conv_w")
                _bn(spec, f"{p}.avd_layer.1", cout // 2)
            cin = cout
            idx += 1
    return [base, base * 4, base * 8, base * 16]


def bf_state_spec(config: Dict) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """Ordered ``name -> (shape, kind)`` of ``BisenetFormer(config).state_dict()``
    (focoos/models/bisenetformer/modelling.py:523-592: BiseNet :235-279 with ContextPath :170-212 / FeatureFusionModule
    :215-237, TransformerDecoder :282-461, PredictionHeads :26-60); pinned against the reference's own key dump in
    tests/golden/bf_l_state_keys.json."""
    bb = config["backbone_config"]
    if bb.get("model_type") != "stdc" or bb.get("block_type", "cat") != "cat":
        raise ValueError("engine state spec covers STDC (cat) backbones (bisenetformer-*)")
    nc = int(config["num_classes"])
    fd = int(config.get("pixel_decoder_feat_dim", 128))
    od = int(config.get("pixel_decoder_out_dim", 128))
    hd = int(config.get("transformer_predictor_hidden_dim", 256))
    md = int(config.get("transformer_predictor_out_dim", 128))
    ffd = int(config.get("transformer_predictor_dim_feedforward", 1024))
    nl = int(config.get("transformer_predictor_dec_layers", 6))
    nq = int(config.get("num_queries", 100))
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    chans = stdc_spec(spec, "pixel_decoder.backbone", int(bb.get("base", 64)), tuple(bb.get("layers", (4, 5, 3))), int(bb.get("block_num", 4)),
                      int(bb.get("in_chans", 3)))
    P = "pixel_decoder"

    def arm(prefix, cin):
        spec[f"{prefix}.proj.weight"] = ((fd, cin, 1, 1), "conv_w")
        _conv_bn(spec, f"{prefix}.conv", fd, fd, 3, "conv", "bn")
        spec[f"{prefix}.conv_atten.weight"] = ((fd, fd, 1, 1), "conv_w")
        _bn(spec, f"{prefix}.bn_atten", fd)

    arm(f"{P}.cp.arm32", chans[3])
    _conv_bn(spec, f"{P}.cp.conv_avg", chans[3], fd, 1, "conv", "bn")
    _conv_bn(spec, f"{P}.cp.conv_head32", fd, fd, 3, "conv", "bn")
    arm(f"{P}.cp.arm16", chans[2])
    _conv_bn(spec, f"{P}.cp.conv_head16", fd, fd, 3, "conv", "bn")
    for name, cin in (("proj1", chans[1]), ("proj2", fd)):
        spec[f"{P}.ffm.{name}.weight"] = ((fd, cin, 1, 1), "conv_w")
        spec[f"{P}.ffm.{name}.bias"] = ((fd,), "lin_b")
    _conv_bn(spec, f"{P}.ffm.convblk", fd, fd, 1, "conv", "bn")
    spec[f"{P}.ffm.conv1.weight"] = ((fd // 4, fd, 1, 1), "conv_w")
    spec[f"{P}.ffm.conv2.weight"] = ((fd, fd // 4, 1, 1), "conv_w")
    _conv_bn(spec, f"{P}.conv_out", fd, od, 3, "conv", "bn")
    spec["head.criterion.empty_weight"] = ((nc + 1,), "buf")
    H = "head.predictor"
    for li in range(nl):
        _mha(spec, f"{H}.transformer_self_attention_layers.{li}.self_attn", hd)
        _ln(spec, f"{H}.transformer_self_attention_layers.{li}.norm", hd)
    for li in range(nl):
        _mha(spec, f"{H}.transformer_cross_attention_layers.{li}.multihead_attn", hd)
        _ln(spec, f"{H}.transformer_cross_attention_layers.{li}.norm", hd)
    for li in range(nl):
        _linear(spec, f"{H}.transformer_ffn_layers.{li}.linear1", hd, ffd)
        _linear(spec, f"{H}.transformer_ffn_layers.{li}.linear2", ffd, hd)
        _ln(spec, f"{H}.transformer_ffn_layers.{li}.norm", hd)
    spec[f"{H}.query_feat.weight"] = ((nq, hd), "emb")
    spec[f"{H}.query_embed.weight"] = ((nq, hd), "emb")
    for i in range(min(2, nl)):
        spec[f"{H}.input_proj.{i}.weight"] = ((hd, od, 1, 1), "conv_w")
        spec[f"{H}.input_proj.{i}.bias"] = ((hd,), "lin_b")
    _ln(spec, f"{H}.forward_prediction_heads.decoder_norm", hd)
    _linear(spec, f"{H}.forward_prediction_heads.classifier", hd, nc + 1)
    for j, (a, b) in enumerate([(hd, hd), (hd, hd), (hd, md)]):
        _linear(spec, f"{H}.forward_prediction_heads.mask_classifier.layers.{j}", a, b)
    return spec


def state_spec(config: Dict, family: str = "fai_detr"):
    if family == "fai_detr":
        return detr_state_spec(config)
    if family == "fai_mf":
        return mf_state_spec(config)
    if family == "bisenetformer":
        return bf_state_spec(config)
    raise ValueError(f"engine has no state spec for model family {family!r}")
