"""Command-line surface of MI-Seg (reference utils/parser.py:5-150): same flags, types and defaults, expressed as tables."""
from argparse import ArgumentParser

_T = "store_true"

MODEL_ARGS = {
    "monai.net": [
        ("--pretrained", dict(type=str, help="path to pre-trained model checkpoint")),
        ("--ckpt_path", dict(type=str, help="path to a training checkpoint to resume")),
        ("--model_name", dict(default="unetr", type=str, help="unet | unetr | swin_unetr | pre_swin_unetr")),
        ("--in_channels", dict(default=1, type=int)), ("--out_channels", dict(default=14, type=int)),
        ("--roi_x", dict(default=96, type=int)), ("--roi_y", dict(default=96, type=int)), ("--roi_z", dict(default=96, type=int)),
        ("--feature_size", dict(default=[16], type=int, nargs="+")),
        ("--hidden_size", dict(default=768, type=int)), ("--mlp_dim", dict(default=3072, type=int)),
        ("--num_heads", dict(default=12, type=int)), ("--pos_embed", dict(default="perceptron", type=str)),
        ("--no_conv_block", dict(action=_T)), ("--no_res_block", dict(action=_T)),
        ("--dropout_rate", dict(default=0.0, type=float)), ("--spatial_dims", dict(default=3, type=int)),
        ("--qkv_bias", dict(action=_T)),
        ("--vit_norm_name", dict(type=str, default="layer")), ("--vit_norm_no_affine", dict(action=_T)),
        ("--encoder_norm_name", dict(type=str, default="instance")), ("--encoder_norm_no_affine", dict(action=_T)),
        ("--decoder_norm_name", dict(type=str, default="instance")), ("--decoder_norm_no_affine", dict(action=_T)),
        ("--num_groups", dict(type=int, default=4)), ("--num_styles", dict(type=int, default=2)),
        ("--dropout_path_rate", dict(default=0.0, type=float)), ("--attn_drop_rate", dict(default=0.0, type=float)),
        ("--depth_swin_block", dict(default=[2], type=int, nargs="+")), ("--use_checkpoint", dict(action=_T)),
        ("--downsample", dict(default="merging", type=str)), ("--no_normalize_swin", dict(action=_T)),
        ("--pre_swin", dict(type=str, default="")),
        ("--num_layers", dict(type=int, default=4)), ("--strides", dict(default=[2, 2, 2], nargs="+", type=int)),
        ("--kernel_size", dict(default=3, nargs="+", type=int)), ("--up_kernel_size", dict(default=3, nargs="+", type=int)),
        ("--num_res_units", dict(default=2, type=int)), ("--activation", dict(default="prelu", type=str)),
        ("--no_bias", dict(action=_T)), ("--adn_ordering", dict(default="NDA", type=str)), ("--freeze_encoder", dict(action=_T)),
    ],
    "loss": [
        ("--criterion", dict(default="dice_focal", type=str)), ("--squared_dice", dict(action=_T)),
        ("--smooth_nr", dict(default=0.0, type=float)), ("--smooth_dr", dict(default=1e-6, type=float)),
        ("--no_include_background", dict(action=_T)),
    ],
    "optimizer": [
        ("--lr", dict(default=1e-4, type=float)), ("--optim_name", dict(default="adamw", type=str)),
        ("--reg_weight", dict(default=1e-5, type=float)), ("--momentum", dict(default=0.99, type=float)),
        ("--scheduler", dict(default="reduce_on_plateau", type=str)), ("--warmup_epochs", dict(default=50, type=int)),
        ("--patience_scheduler", dict(default=3, type=int)), ("--t_max", dict(default=200, type=int)),
        ("--cycles", dict(default=0.5, type=float)),
    ],
    "inference": [
        ("--infer_overlap", dict(default=0.5, type=float)), ("--sw_batch_size", dict(default=1, type=int)), ("--infer_cpu", dict(action=_T)),
    ],
    "early_stop": [("--patience", dict(default=6, type=int)), ("--min_delta", dict(default=0.001, type=float))],
    "checkpointing": [("--save_top_k", dict(default=3, type=int))],
    "wandb_logger": [
        ("--experiment_name", dict(type=str)), ("--group", dict(type=str)), ("--project", dict(type=str)), ("--entity", dict(type=str)),
        ("--wandb_mode", dict(type=str, default="online")), ("--source", dict(type=int)), ("--alpha_reversal", dict(type=float, default=1.0)),
    ],
}

DATA_ARGS = {
    "dataset(s)": [
        ("--data_dirs", dict(default=["dataset/MM-WHS", "dataset/MM-WHS"], type=str, nargs="+")),
        ("--json_lists", dict(default=["CT_fold1.json", "MR.json"], nargs="+", type=str)),
        ("--space_x", dict(default=1.0, type=float)), ("--space_y", dict(default=1.0, type=float)), ("--space_z", dict(default=1.0, type=float)),
        ("--patches_training_sample", dict(default=1, type=int)),
        ("--randFlipd_prob", dict(default=0.2, type=float)), ("--randRotate90d_prob", dict(default=0.2, type=float)),
        ("--randScaleIntensityd_prob", dict(default=0.1, type=float)), ("--randShiftIntensityd_prob", dict(default=0.1, type=float)),
        ("--use_normal_dataset", dict(action=_T)), ("--cache_num", dict(default=24, type=int)), ("--loader_workers", dict(default=8, type=int)),
        ("--batch_size", dict(default=1, type=int)), ("--num_workers", dict(default=8, type=int)),
    ],
}

TUNE_ARGS = {
    "tune": [
        ("--study_name", dict(default="experiment", type=str)), ("--n_trials", dict(type=int)), ("--timeout", dict(type=int)),
        ("--max_epochs", dict(default=2, type=int)), ("--check_val_every_n_epoch", dict(default=1, type=int)),
        ("--no_gpu", dict(action=_T)), ("--no_amp", dict(action=_T)), ("--iters_to_accumulate", dict(default=1, type=int)),
        ("--default_root_dir", dict(default="./experiments", type=str)), ("--port", dict(default="23456", type=str)),
        ("--storage_name", dict(default="MI-Seg", type=str)), ("--min_lr", dict(default=1e-5, type=float)), ("--max_lr", dict(default=5e-3, type=float)),
    ],
}


def _add(parser: ArgumentParser, table):
    for group_name, entries in table.items():
        grp = parser.add_argument_group(group_name)
        for flag, kw in entries:
            grp.add_argument(flag, **kw)
    return parser


def add_model_argparse_args(parser: ArgumentParser):
    return _add(parser, MODEL_ARGS)


def add_data_argparse_args(parser: ArgumentParser):
    return _add(parser, DATA_ARGS)


def add_tune_argparse_args(parser: ArgumentParser):
    return _add(parser, TUNE_ARGS)
