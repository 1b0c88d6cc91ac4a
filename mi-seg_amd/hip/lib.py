"""ctypes binding of csrc/libmiseg_hip.so (the C ABI declared in include/miseg_hip.h).

The product path has NO fallback: if the shared object is missing or a symbol is absent, importing an op raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISEG_HIP_LIB") or os.path.join(os.path.dirname(HERE), "csrc", "libmiseg_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_LEAKY, ACT_GELU, ACT_PRELU = 0, 1, 2, 3
MAX_STYLES = 4

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
fp4 = vp * MAX_STYLES


class _S(C.Structure):
    pass


ABI_VERSION = 9          # == MISEG_ABI_VERSION of include/miseg_hip.h; load() refuses a library that reports another
C_NAMES = {}             # ctypes mirror -> name of the C struct it mirrors (tests/test_abi.py checks sizeof / offsetof of every field)


def _struct(name, fields, cname=None):
    t = type(name, (C.Structure,), {"_fields_": fields})
    C_NAMES[t] = cname
    return t


InstnormStats = _struct("InstnormStats", cname="miseg_instnorm_stats_params", fields=[("x", vp), ("ldx", i64), ("B", i32), ("S", i32), ("C", i32), ("dtype", i32), ("stat", vp)])
InstnormApply = _struct("InstnormApply", cname="miseg_instnorm_apply_params", fields=[("x", vp), ("ldx", i64), ("res", vp), ("ldres", i64), ("y", vp), ("ldy", i64),
                                          ("B", i32), ("S", i32), ("C", i32), ("dtype", i32), ("stat", vp), ("eps", f32),
                                          ("styles", vp), ("num_styles", i32), ("gamma", fp4), ("beta", fp4),
                                          ("act", i32), ("slope", f32), ("res_stat", vp), ("res_gamma", fp4), ("res_beta", fp4),
                                          ("r1x", vp), ("ldr1x", i64), ("r1w", vp)])
InstnormBwd = _struct("InstnormBwd", cname="miseg_instnorm_bwd_params", fields=[("dy", vp), ("lddy", i64), ("y", vp), ("ldy", i64), ("x", vp), ("ldx", i64),
                                      ("dx", vp), ("lddx", i64), ("dres", vp), ("lddres", i64),
                                      ("B", i32), ("S", i32), ("C", i32), ("dtype", i32), ("stat", vp), ("eps", f32), ("dstat", vp),
                                      ("styles", vp), ("num_styles", i32), ("gamma", fp4), ("dgamma", fp4), ("dbeta", fp4),
                                      ("act", i32), ("slope", f32), ("gadd", vp), ("ldgadd", i64), ("beta", fp4)])
InstnormPairBwd = _struct("InstnormPairBwd", cname="miseg_instnorm_pair_bwd_params", fields=[("dy", vp), ("lddy", i64), ("y", vp), ("ldy", i64), ("xa", vp), ("ldxa", i64), ("xb", vp), ("ldxb", i64),
                                              ("dxa", vp), ("lddxa", i64), ("dxb", vp), ("lddxb", i64), ("B", i32), ("S", i32), ("C", i32), ("dtype", i32),
                                              ("stat_a", vp), ("stat_b", vp), ("eps", f32), ("dstat_a", vp), ("dstat_b", vp), ("styles", vp),
                                              ("num_styles", i32), ("gamma_a", fp4), ("gamma_b", fp4), ("dgamma_a", fp4), ("dbeta_a", fp4),
                                              ("dgamma_b", fp4), ("dbeta_b", fp4), ("slope", f32), ("beta_a", fp4), ("beta_b", fp4),
                                              ("r1x", vp), ("ldr1x", i64), ("r1w", vp), ("r1dw", vp)])
LayernormFwd = _struct("LayernormFwd", cname="miseg_layernorm_fwd_params", fields=[("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("rows", i64), ("C", i32),
                                        ("dtype", i32), ("eps", f32), ("gamma", vp), ("beta", vp), ("mean", vp), ("rstd", vp)])
LayernormBwd = _struct("LayernormBwd", cname="miseg_layernorm_bwd_params", fields=[("dy", vp), ("lddy", i64), ("x", vp), ("ldx", i64), ("dx", vp), ("lddx", i64),
                                        ("rows", i64), ("C", i32), ("dtype", i32), ("gamma", vp), ("mean", vp), ("rstd", vp),
                                        ("dgamma", vp), ("dbeta", vp)])
NormRef = _struct("NormRef", cname="miseg_norm_ref", fields=[("stat", vp), ("styles", vp), ("num_styles", i32), ("eps", f32), ("gamma", fp4), ("beta", fp4)])
Gemm = _struct("Gemm", cname="miseg_gemm_params", fields=[("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64), ("M", i32), ("N", i32),
                        ("K", i32), ("ta", i32), ("tb", i32), ("dtype", i32), ("out_dtype", i32), ("bias", vp), ("act", i32),
                        ("accumulate", i32), ("split_k", i32), ("workspace", vp), ("res", vp), ("ldres", i64), ("aux", vp), ("ldaux", i64),
                        ("epi_mode", i32), ("defer_reduce", i32), ("stat", vp), ("scat_d", i32), ("scat_h", i32), ("scat_w", i32),
                        ("scat_cout", i32), ("an", NormRef), ("an_out", vp), ("ld_an_out", i64), ("stat_mode", i32), ("bs_x", vp), ("ld_bs_x", i64),
                        ("bs_stat", vp), ("bs_eps", f32), ("tn_colsum", vp)])
TnReduceDesc = _struct("TnReduceDesc", cname="miseg_tn_reduce_desc", fields=[("partial", vp), ("C", vp), ("ldc", i64), ("M", i32), ("N", i32), ("splits", i32), ("block0", i32),
                                                                            ("regroup", i32), ("pad_", i32)])
ColsumDesc = _struct("ColsumDesc", cname="miseg_colsum_desc", fields=[("x", vp), ("ldx", i64), ("rows", i64), ("out", vp), ("C", i32), ("block0", i32)])
GemmTnDesc = _struct("GemmTnDesc", cname="miseg_gemm_tn_desc", fields=[("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64), ("M", i32), ("N", i32), ("K", i32), ("zeroed", i32),
                                                                      ("regroup", i32), ("pad_", i32)])
Colsum = _struct("Colsum", cname="miseg_colsum_params", fields=[("x", vp), ("ldx", i64), ("rows", i64), ("C", i32), ("dtype", i32), ("out", vp), ("accumulate", i32)])
Conv3 = _struct("Conv3", cname="miseg_conv3_params", fields=[("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("wpk", vp), ("B", i32), ("D", i32), ("H", i32),
                          ("W", i32), ("Cin", i32), ("Cout", i32), ("dtype", i32), ("workspace", vp), ("res", vp), ("ldres", i64), ("stat", vp), ("background", i32), ("defer_slabs", i32),
                          ("sc_x", vp), ("ld_sc_x", i64), ("sc_w", vp), ("sc_C", i32), ("s2c_out", vp), ("s2c_C", i32),
                          ("fs_w", vp), ("fs_y", vp), ("ld_fs_y", i64), ("fs_stat", vp)])
PackConv3Desc = _struct("PackConv3Desc", cname="miseg_pack_conv3_desc", fields=[("w", vp), ("fwd_pack", vp), ("bwd_pack", vp), ("Cin", i32), ("Cout", i32), ("tile0", i32), ("pad_", i32)])
PackConv3 = _struct("PackConv3", cname="miseg_pack_conv3_params", fields=[("w", vp), ("fwd_pack", vp), ("bwd_pack", vp), ("Cin", i32), ("Cout", i32), ("dtype", i32)])
Conv3Wgrad = _struct("Conv3Wgrad", cname="miseg_conv3_wgrad_params", fields=[("x", vp), ("ldx", i64), ("dy", vp), ("lddy", i64), ("dw", vp), ("B", i32), ("D", i32),
                                    ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32), ("dtype", i32), ("accumulate", i32),
                                    ("workspace", vp), ("max_workgroups", i32)])
Winattn = _struct("Winattn", cname="miseg_winattn_params", fields=[("qkv", vp), ("ldq", i64), ("out", vp), ("ldo", i64), ("qkv_bias", vp), ("bias_table", vp),
                              ("lse", vp), ("B", i32), ("D", i32), ("H", i32), ("W", i32), ("C", i32), ("heads", i32),
                              ("dtype", i32), ("wd", i32), ("wh", i32), ("ww", i32), ("sd", i32), ("sh", i32), ("sw", i32),
                              ("tw", i32), ("scale", f32), ("drop_p", f32), ("drop_seed", C.c_uint64), ("drop_stream", C.c_uint64), ("drop_step_dev", vp)])
WinattnBwd = _struct("WinattnBwd", cname="miseg_winattn_bwd_params", fields=[("f", Winattn), ("dout", vp), ("lddo", i64), ("dqkv", vp), ("lddq", i64),
                                    ("dqkv_bias", vp), ("dbias_table", vp)])
Add = _struct("Add", cname="miseg_add_params", fields=[("a", vp), ("lda", i64), ("b", vp), ("ldb", i64), ("y", vp), ("ldy", i64), ("rows", i64), ("C", i32), ("dtype", i32)])
GraphSplitInfo = _struct("GraphSplitInfo", cname="miseg_graph_split_info", fields=[("nodes", i32), ("lanes", i32), ("segments", i32), ("crossing_edges", i32),
                                                                                  ("side_streams", i32), ("main_lane_nodes", i32), ("streams_concurrent", i32)])
Affine2 = _struct("Affine2", cname="miseg_affine2_params", fields=[("struct_size", C.c_uint32), ("a", vp), ("lda", i64), ("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("coef", vp),
                                                                  ("B", i32), ("S", i32), ("C", i32), ("dtype", i32)])
Copy2d = _struct("Copy2d", cname="miseg_copy2d_params", fields=[("src", vp), ("lds", i64), ("sdtype", i32), ("dst", vp), ("ldd", i64), ("ddtype", i32), ("rows", i64), ("C", i32)])
CastDesc = _struct("CastDesc", cname="miseg_cast_desc", fields=[("src", vp), ("dst", vp), ("R", i32), ("C", i32), ("transpose", i32), ("inner", i32), ("outer", i32), ("tile0", i32)])
Resample2 = _struct("Resample2", cname="miseg_resample2_params", fields=[("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("B", i32), ("D", i32), ("H", i32), ("W", i32), ("C", i32),
                                  ("dtype", i32), ("dir", i32)])
Rowbias = _struct("Rowbias", cname="miseg_rowbias_params", fields=[("x", vp), ("ldx", i64), ("bias", vp), ("y", vp), ("ldy", i64), ("rows", i64), ("C", i32), ("dtype", i32)])
PreluFwd = _struct("PreluFwd", cname="miseg_prelu_fwd_params", fields=[("x", vp), ("ldx", i64), ("slope", vp), ("y", vp), ("ldy", i64), ("rows", i64), ("C", i32), ("dtype", i32)])
PreluBwd = _struct("PreluBwd", cname="miseg_prelu_bwd_params", fields=[("dy", vp), ("lddy", i64), ("x", vp), ("ldx", i64), ("slope", vp), ("dx", vp), ("lddx", i64), ("dslope", vp),
                                ("rows", i64), ("C", i32), ("dtype", i32), ("scratch", vp)])
Cast = _struct("Cast", cname="miseg_cast_params", fields=[("src", vp), ("dst", vp), ("R", i32), ("C", i32), ("dtype", i32), ("transpose", i32)])
GeluFwd = _struct("GeluFwd", cname="miseg_gelu_fwd_params", fields=[("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("rows", i64), ("C", i32), ("dtype", i32)])
GeluBwd = _struct("GeluBwd", cname="miseg_gelu_bwd_params", fields=[("dy", vp), ("lddy", i64), ("x", vp), ("ldx", i64), ("dx", vp), ("lddx", i64), ("rows", i64), ("C", i32), ("dtype", i32)])
S2C = _struct("S2C", cname="miseg_s2c_params", fields=[("src", vp), ("lds", i64), ("dst", vp), ("ldd", i64), ("B", i32), ("D", i32), ("H", i32), ("W", i32),
                      ("C", i32), ("dtype", i32), ("offsets", C.c_int8 * 24)])
PatchEmbed = _struct("PatchEmbed", cname="miseg_patch_embed_params", fields=[("x", vp), ("y", vp), ("ldy", i64), ("w", vp), ("bias", vp), ("B", i32), ("Cin", i32), ("D", i32),
                                    ("H", i32), ("W", i32), ("Cout", i32), ("dtype", i32)])
PatchEmbedBwd = _struct("PatchEmbedBwd", cname="miseg_patch_embed_bwd_params", fields=[("x", vp), ("dy", vp), ("lddy", i64), ("dw", vp), ("dbias", vp), ("B", i32), ("Cin", i32),
                                          ("D", i32), ("H", i32), ("W", i32), ("Cout", i32), ("dtype", i32), ("workspace", vp)])
Conv3Thin = _struct("Conv3Thin", cname="miseg_conv3_thin_params", fields=[("x", vp), ("y", vp), ("ldy", i64), ("w", vp), ("B", i32), ("Cin", i32), ("D", i32), ("H", i32),
                                  ("W", i32), ("Cout", i32), ("dtype", i32)])
Conv3ThinWgrad = _struct("Conv3ThinWgrad", cname="miseg_conv3_thin_wgrad_params", fields=[("x", vp), ("dy", vp), ("lddy", i64), ("dw", vp), ("B", i32), ("Cin", i32), ("D", i32),
                                            ("H", i32), ("W", i32), ("Cout", i32), ("dtype", i32), ("workspace", vp)])
Head = _struct("Head", cname="miseg_head_params", fields=[("x", vp), ("ldx", i64), ("y", vp), ("w", vp), ("bias", vp), ("B", i32), ("S", i32), ("Cin", i32),
                        ("Cout", i32), ("dtype", i32)])
HeadBwd = _struct("HeadBwd", cname="miseg_head_bwd_params", fields=[("x", vp), ("ldx", i64), ("dy", vp), ("dx", vp), ("lddx", i64), ("w", vp), ("dw", vp), ("dbias", vp),
                              ("B", i32), ("S", i32), ("Cin", i32), ("Cout", i32), ("dtype", i32)])
Im2col3 = _struct("Im2col3", cname="miseg_im2col3_params", fields=[("src", vp), ("lds", i64), ("dst", vp), ("ldd", i64), ("B", i32), ("D", i32), ("H", i32), ("W", i32),
                              ("C", i32), ("dtype", i32)])

u32, u64p = C.c_uint32, C.POINTER(C.c_uint64)
Mlp = _struct("Mlp", cname="miseg_mlp_params", fields=[("struct_size", u32), ("M", i32), ("C", i32), ("HID", i32), ("dtype", i32), ("x", vp), ("ldx", i64),
                      ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("res", vp), ("ldres", i64), ("y", vp), ("ldy", i64), ("stat", vp),
                      ("dy", vp), ("lddy", i64), ("w2t", vp), ("w1t", vp), ("dz", vp), ("lddz", i64), ("h", vp), ("ldh", i64), ("dx", vp), ("lddx", i64),
                      ("an", NormRef), ("an_out", vp), ("ld_an_out", i64), ("bs_x", vp), ("ld_bs_x", i64), ("bs_stat", vp), ("bs_eps", f32), ("bs_dstat", vp)])
SegLoss = _struct("SegLoss", cname="miseg_seg_loss_params", fields=[
    ("struct_size", u32), ("kind", i32), ("logits", vp), ("label", vp), ("label_dtype", i32), ("B", i32), ("C", i32), ("S", i64),
    ("include_background", i32), ("squared_pred", i32), ("smooth_nr", f32), ("smooth_dr", f32), ("gamma", f32), ("lambda_dice", f32),
    ("lambda_other", f32), ("workspace", vp), ("sums", vp), ("loss", vp), ("gscale", vp), ("dlogits", vp)])
DiceMetric = _struct("DiceMetric", cname="miseg_dice_metric_params", fields=[
    ("struct_size", u32), ("logits", vp), ("label", vp), ("label_dtype", i32), ("B", i32), ("C", i32), ("S", i64), ("counts", vp), ("dice", vp)])
OptDesc = _struct("OptDesc", cname="miseg_opt_desc", fields=[("param", vp), ("off", i64), ("n", i32), ("block0", i32)])
OptPackMap = _struct("OptPackMap", cname="miseg_opt_pack_map", fields=[("off", i64), ("param_index", i32), ("pad_", i32)])
OptStep = _struct("OptStep", cname="miseg_opt_step_params", fields=[
    ("struct_size", u32), ("kind", i32), ("descs_dev", vp), ("ndesc", i32), ("total_blocks", i32), ("grad", vp), ("state1", vp), ("state2", vp),
    ("used", vp), ("steps", vp), ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("momentum", f32), ("lr_dev", vp), ("params_version", vp),
    ("index", vp), ("count_n", i32)])
Stitch = _struct("Stitch", cname="miseg_stitch_params", fields=[
    ("struct_size", u32), ("win", vp), ("out", vp), ("count", vp), ("C", i32), ("D", i32), ("H", i32), ("W", i32), ("rd", i32), ("rh", i32), ("rw", i32),
    ("nd", i32), ("nh", i32), ("nw", i32), ("start_d", vp), ("start_h", vp), ("start_w", vp), ("d_begin", i32), ("d_count", i32)])
AugSample = _struct("AugSample", cname="miseg_aug_sample", fields=[("origin", i32 * 3), ("flip", i32 * 3), ("rot_k", i32), ("scale", f32), ("shift", f32)])
Augment = _struct("Augment", cname="miseg_augment_params", fields=[
    ("struct_size", u32), ("image", vp), ("label", vp), ("label_bytes", i32), ("C", i32), ("D", i32), ("H", i32), ("W", i32), ("rd", i32), ("rh", i32),
    ("rw", i32), ("n", i32), ("out_image", vp), ("out_label", vp), ("samples_host", vp)])
Resample3d = _struct("Resample3d", cname="miseg_resample3d_params", fields=[
    ("struct_size", u32), ("in", vp), ("out", vp), ("C", i32), ("Di", i32), ("Hi", i32), ("Wi", i32), ("Do", i32), ("Ho", i32), ("Wo", i32),
    ("mode", i32), ("elem_bytes", i32)])
Dropout = _struct("Dropout", cname="miseg_dropout_params", fields=[
    ("struct_size", u32), ("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("rows", i64), ("C", i32), ("dtype", i32), ("rows_per_sample", i64),
    ("p", f32), ("seed", C.c_uint64), ("stream_id", C.c_uint64), ("step_dev", vp)])
AUG_MAX_SAMPLES = 16
LABEL_F32, LABEL_I32, LABEL_I64, LABEL_U8 = 0, 1, 2, 3
LOSS_DICE_FOCAL, LOSS_DICE_CE = 0, 1
OPT_ADAMW, OPT_ADAM, OPT_SGD_NESTEROV = 0, 1, 2
OPT_BLOCK = 4096
STITCH_MAX_WINDOWS = 64
FILL_RANGES = 16

# symbol -> (restype, argtypes); every prototype of include/miseg_hip.h appears here (checked by tests/test_abi.py)
PROTOS = {
    "miseg_abi_version": (i32, []),
    "miseg_last_error": (C.c_char_p, []),
    "miseg_device_arch": (i32, [C.c_char_p, C.c_size_t]),
    "miseg_instnorm_stat_bytes": (C.c_size_t, [i32, i32]),
    "miseg_instnorm_stats": (i32, [C.POINTER(InstnormStats), vp]),
    "miseg_instnorm_apply": (i32, [C.POINTER(InstnormApply), vp]),
    "miseg_instnorm_bwd_slabs": (i32, [C.POINTER(InstnormBwd), vp, i32, i64, vp]),
    "miseg_instnorm_fwd_slabs": (i32, [C.POINTER(InstnormApply), vp, i32, i64, vp]),
    "miseg_instnorm_fused_max_rows": (i32, []),
    "miseg_conv3_fwd_splits": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "miseg_conv3_fuses_shortcut": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "miseg_conv3_fwd_tiny": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "miseg_conv3_fuses_s2c": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "miseg_conv3_fuses_fwd_shortcut": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "miseg_instnorm_fwd": (i32, [C.POINTER(InstnormApply), vp]),
    "miseg_instnorm_bwd": (i32, [C.POINTER(InstnormBwd), vp]),
    "miseg_instnorm_pair_bwd": (i32, [C.POINTER(InstnormPairBwd), vp]),
    "miseg_layernorm_fwd": (i32, [C.POINTER(LayernormFwd), vp]),
    "miseg_layernorm_bwd": (i32, [C.POINTER(LayernormBwd), vp]),
    "miseg_gemm_fuses_stat": (i32, [C.POINTER(Gemm)]),
    "miseg_gemm_fuses_scatter": (i32, [C.POINTER(Gemm)]),
    "miseg_gemm_fuses_anorm": (i32, [C.POINTER(Gemm)]),
    "miseg_gemm_fuses_bstat": (i32, [C.POINTER(Gemm)]),
    "miseg_instnorm_bwd_apply": (i32, [C.POINTER(InstnormBwd), vp]),
    "miseg_rank1_stats": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, vp]),
    "miseg_mlp_fused": (i32, [i32, i32, i32, i32]),
    "miseg_mlp_fwd": (i32, [C.POINTER(Mlp), vp]),
    "miseg_mlp_bwd": (i32, [C.POINTER(Mlp), vp]),
    "miseg_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(Gemm)]),
    "miseg_gemm": (i32, [C.POINTER(Gemm), vp]),
    "miseg_permute3": (i32, [vp, vp, i32, i32, i32, i64, i64, i64, i32, vp]),
    "miseg_colsum": (i32, [C.POINTER(Colsum), vp]),
    "miseg_gemm_tn_splits": (i32, [C.POINTER(Gemm)]),
    "miseg_gemm_tn_fuses_colsum": (i32, [C.POINTER(Gemm)]),
    "miseg_gemm_tn_reduce_batch": (i32, [vp, i32, vp]),
    "miseg_gemm_tn_group": (i32, [vp, i32, i32, vp]),
    "miseg_colsum_batch": (i32, [vp, i32, i32, vp]),
    "miseg_conv3_fwd_workspace_bytes": (C.c_size_t, [i32, i32, i32, i32, i32, i32, i32]),
    "miseg_conv3_fwd": (i32, [C.POINTER(Conv3), vp]),
    "miseg_pack_conv3_elems": (C.c_size_t, [i32, i32, i32, i32]),
    "miseg_conv3_k96": (i32, [i32, i32]),
    "miseg_pack_conv3_tiles": (i32, [i32, i32, i32]),
    "miseg_pack_conv3_weight": (i32, [C.POINTER(PackConv3), vp]),
    "miseg_pack_conv3_batch": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "miseg_conv3_wgrad_workspace_bytes": (C.c_size_t, [i32, i32, i32, i32, i32, i32]),
    "miseg_conv3_wgrad": (i32, [C.POINTER(Conv3Wgrad), vp]),
    "miseg_conv3_wgrad_group_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3Wgrad), i32]),
    "miseg_conv3_wgrad_group": (i32, [C.POINTER(Conv3Wgrad), i32, vp, vp]),
    "miseg_winattn_fwd": (i32, [C.POINTER(Winattn), vp]),
    "miseg_winattn_on_matrix_cores": (i32, [vp]),
    "miseg_winattn_bwd": (i32, [C.POINTER(WinattnBwd), vp]),
    "miseg_add": (i32, [C.POINTER(Add), vp]),
    "miseg_affine2": (i32, [C.POINTER(Affine2), vp]),
    "miseg_instnorm_bwd_reduce": (i32, [C.POINTER(InstnormBwd), vp]),
    "miseg_copy2d": (i32, [C.POINTER(Copy2d), vp]),
    "miseg_cast_matrix": (i32, [C.POINTER(Cast), vp]),
    "miseg_gelu_fwd": (i32, [C.POINTER(GeluFwd), vp]),
    "miseg_gelu_bwd": (i32, [C.POINTER(GeluBwd), vp]),
    "miseg_space_to_channel": (i32, [C.POINTER(S2C), vp]),
    "miseg_channel_to_space": (i32, [C.POINTER(S2C), vp]),
    "miseg_patch_embed_fwd": (i32, [C.POINTER(PatchEmbed), vp]),
    "miseg_patch_embed_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(PatchEmbedBwd)]),
    "miseg_patch_embed_bwd": (i32, [C.POINTER(PatchEmbedBwd), vp]),
    "miseg_conv3_thin_fwd": (i32, [C.POINTER(Conv3Thin), vp]),
    "miseg_conv3_thin_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3ThinWgrad)]),
    "miseg_conv3_thin_wgrad": (i32, [C.POINTER(Conv3ThinWgrad), vp]),
    "miseg_head_fwd": (i32, [C.POINTER(Head), vp]),
    "miseg_head_bwd": (i32, [C.POINTER(HeadBwd), vp]),
    "miseg_im2col3": (i32, [C.POINTER(Im2col3), vp]),
    "miseg_col2im3": (i32, [C.POINTER(Im2col3), vp]),
    "miseg_fill32": (i32, [vp, C.c_uint32, C.c_size_t, vp]),
    "miseg_fill32_ranges": (i32, [vp, C.c_uint32, vp, i32, vp]),
    "miseg_param_cast_batch": (i32, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "miseg_resample2": (i32, [C.POINTER(Resample2), vp]),
    "miseg_rowbias_add": (i32, [C.POINTER(Rowbias), vp]),
    "miseg_prelu_fwd": (i32, [C.POINTER(PreluFwd), vp]),
    "miseg_prelu_bwd": (i32, [C.POINTER(PreluBwd), vp]),
    "miseg_layout_ncdhw": (i32, [vp, i64, vp, i32, i32, i64, i32, i32, vp]),
    "miseg_ncdhw_to_rows": (i32, [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, vp]),
    "miseg_seg_loss_workspace_bytes": (C.c_size_t, [i32, i32, i64]),
    "miseg_seg_loss_fwd": (i32, [C.POINTER(SegLoss), vp]),
    "miseg_seg_loss_bwd": (i32, [C.POINTER(SegLoss), vp]),
    "miseg_dice_metric": (i32, [C.POINTER(DiceMetric), vp]),
    "miseg_opt_step": (i32, [C.POINTER(OptStep), vp]),
    "miseg_opt_step_pack_conv3": (i32, [C.POINTER(OptStep), vp, vp, i32, i32, i32, vp, vp]),
    "miseg_stitch_windows": (i32, [C.POINTER(Stitch), vp]),
    "miseg_augment_crop": (i32, [C.POINTER(Augment), vp]),
    "miseg_resample3d": (i32, [C.POINTER(Resample3d), vp]),
    "miseg_dropout": (i32, [C.POINTER(Dropout), vp]),
    "miseg_counter_add": (i32, [vp, C.c_uint64, vp]),
    "miseg_conv3_wgrad_tiny": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "miseg_counter_copy": (i32, [vp, vp, vp]),
    "miseg_abi_struct_size": (C.c_size_t, [C.c_char_p]),
    "miseg_device_check": (i32, [i32]),
    "miseg_source_digest": (C.c_char_p, []),
}
# measurement / experiment entry points of include/miseg_hip_debug.h (same shared object; no product path calls them)
DEBUG_PROTOS = {
    "miseg_debug_stamp": (i32, [vp, vp]),
    "miseg_prof_arm": (i32, [i32]),
    "miseg_prof_read": (i32, [vp, vp, i32]),
    "miseg_prof_available": (i32, []),
    "miseg_flag_wait": (i32, [vp, vp, C.c_uint64, vp, vp]),
    "miseg_streams_run_concurrently": (i32, [vp, vp]),
    "miseg_graph_split_create": (i32, [vp, vp, i32, C.POINTER(vp), vp]),
    "miseg_graph_split_launch": (i32, [vp, vp]),
    "miseg_graph_split_destroy": (None, [vp]),
}
DEBUG_STRUCTS = {"miseg_graph_split_info"}      # mirrors of structs that include/miseg_hip_debug.h declares
PROF_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libmiseg_hip_prof.so")

_lib = None


class MisegHipError(RuntimeError):
    pass


def source_digest():
    """sha256 over the library's sources as csrc/build.py hashes them (None when the sources are not beside the library)"""
    try:
        from ..csrc import build as _b
        return _b.source_digest()
    except Exception:
        return None


def _open(path):
    if not os.path.exists(path):
        raise MisegHipError(f"{path} is missing: run `python __graft_entry__.py` (build()) first; "
                            "the MI355X path has no fallback implementation")
    lib = C.CDLL(path)
    for table in (PROTOS, DEBUG_PROTOS):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
    # header / library / binding drift is an error at load time, not a kernel reading past a struct later
    if lib.miseg_abi_version() != ABI_VERSION:
        raise MisegHipError(f"{path} reports ABI version {lib.miseg_abi_version()}, this binding mirrors version {ABI_VERSION}: rebuild (build())")
    for t, cname in C_NAMES.items():
        if cname is not None and lib.miseg_abi_struct_size(cname.encode()) != C.sizeof(t):
            raise MisegHipError(f"sizeof({cname}) is {lib.miseg_abi_struct_size(cname.encode())} in the library, {C.sizeof(t)} in hip/lib.py")
    # a library older than the sources beside it is refused (the in-tree .so travels with its sources; MISEG_ALLOW_STALE_LIB=1 for A/B runs
    # against a kept build)
    want = None if (os.environ.get("MISEG_ALLOW_STALE_LIB") or os.environ.get("MISEG_HIP_LIB")) else source_digest()
    have = lib.miseg_source_digest().decode()
    if want is not None and have != want:
        raise MisegHipError(f"{path} was built from other sources (digest {have[:12]}, the tree has {want[:12]}): run build()")
    return lib


def load():
    """Load the shared object (once).  Raises if it was not built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


class profiling_library:
    """`with profiling_library():` - every call of the block goes to libmiseg_hip_prof.so, the measurement build of the SAME objects linked
    with -Wl,--wrap=hipLaunchKernel (csrc/build.py, csrc/common.cpp): bench.py's roofline leg times one extra eager step through it.  The
    library keeps no state of the caller's (buffers, packs, pools are the caller's), so the two builds are interchangeable call by call."""
    _prof = None

    def __enter__(self):
        global _lib
        load()
        if profiling_library._prof is None:
            profiling_library._prof = _open(PROF_LIB_PATH)
        self.keep, _lib = _lib, profiling_library._prof
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.keep
        return False


def check_device(index=0):
    """raise unless HIP device `index` is the architecture the library's code objects were built for"""
    check(load().miseg_device_check(index), "device_check")


def check(rc, what=""):
    if rc != 0:
        msg = load().miseg_last_error().decode()
        if rc == -1:
            raise ValueError(f"miseg {what}: {msg}")
        raise MisegHipError(f"miseg {what} failed ({rc}): {msg}")
