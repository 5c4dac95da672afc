"""Flag surface of the hot path: mirror of the reference's ``scripts/config.py:5-82``.

``get_args()`` accepts every flag the reference accepts, with the same names, types and defaults,
so an ``opt`` namespace built here is interchangeable with one built by the reference.  The build
adds three flags ("Config / flags"):

* ``--ode_solver`` additionally accepts ``rk4`` (alias ``runge_kutta``; 3/8-rule, the step
  torchdiffeq's ``method="rk4"`` takes) and ``rk4_classic`` (1/6-1/3-1/3-1/6),
* ``--ode_substeps``: equal fixed sub-steps per observation interval for the fixed-step solvers,
* ``--dtype``: arithmetic type of the HIP path (only ``fp32`` carries the 1e-4 parity claim).

The table below is data, not code copied from the reference: (name, type, default, help, extra).
"""
import argparse

_STORE_TRUE = "store_true"

# fmt: off
_FLAGS = [
    # paths
    ("data_dir", str, "/mnt/data0/marco/KITTI/data", "path to the dataset", {}),
    ("gpu_ids", str, "0", "gpu ids: e.g. 0  0,1,2, 0,2. use -1 for CPU", {}),
    ("save_dir", str, "./results", "path to save the result", {}),
    ("plot_dir", str, "./results", "path to save the log", {}),
    # run bookkeeping (unused by the hot path, kept so reference command lines parse)
    ("experiment_name", str, "experiment", "experiment name", {}),
    ("wandb", _STORE_TRUE, False, "whether to use wandb logging", {}),
    ("wandb_group", str, "ode-rnn", "group of the wandb run", {}),
    ("sweep", _STORE_TRUE, False, "whether to use wandb sweep", {}),
    ("resume", str, None, "resume training (wandb run id)", {}),
    ("pretrain_flownet", str, "./pretrained_models/flownets_bn_EPE2.459.pth.tar", "pre-trained flownet checkpoint", {}),
    ("pretrain", str, None, "path to the pretrained model", {}),
    ("train_seq", str, ["00", "01", "02", "04", "08", "09"], "sequences for training", {"nargs": "+"}),
    ("val_seq", str, ["06"], "sequences for validation", {"nargs": "+"}),
    ("seed", int, 0, "random seed", {}),
    ("workers", int, 8, "number of workers in dataloader", {}),
    ("print_frequency", int, 10, "print frequency for loss values", {}),
    # training hyper-parameters (out of scope for the forward hot path; parsed for compatibility)
    ("model_type", str, "ode-rnn", "type of model [ode-rnn (ODE-VIO), cde, rnn]", {}),
    ("optimizer", str, "Adam", "type of optimizer [Adam, SGD]", {}),
    ("grad_accumulation_steps", int, 1, "gradient accumulation steps before updating", {}),
    ("freeze_encoder", _STORE_TRUE, False, "freeze the encoder or not", {}),
    ("weight_decay", float, 5e-5, "weight decay for the optimizer", {}),
    ("batch_size", int, 26, "batch size", {}),
    ("shuffle", bool, True, "shuffle data samples or not", {}),
    ("epochs_warmup", int, 20, "number of epochs for warmup", {}),
    ("epochs_joint", int, 40, "number of epochs for joint training", {}),
    ("epochs_fine", int, 40, "number of epochs for finetuning", {}),
    ("lr_warmup", float, 1e-4, "learning rate for warming up stage", {}),
    ("lr_joint", float, 1e-5, "learning rate for joint training stage", {}),
    ("lr_fine", float, 1e-6, "learning rate for finetuning stage", {}),
    ("gradient_clip", float, 5, "gradient clipping norm/clip value", {}),
    # data
    ("data_dropout", float, 0.0, "irregularity in the dataset by dropping out randomly", {}),
    ("data_dropout_std", float, 0.0, "std of irregularity across each epoch", {}),
    ("eval_data_dropout", float, 0.0, "irregularity in the eval dataset", {}),
    ("img_w", int, 512, "image width", {}),
    ("img_h", int, 256, "image height", {}),
    ("v_f_len", int, 512, "visual feature length", {}),
    ("i_f_len", int, 256, "imu feature length", {}),
    ("imu_dropout", float, 0, "dropout for the IMU encoder", {}),
    ("hflip", _STORE_TRUE, False, "whether to use horizontal flipping as augmentation", {}),
    ("color", _STORE_TRUE, False, "whether to use color augmentations", {}),
    ("seq_len", int, 11, "sequence length of images", {}),
    ("normalize", _STORE_TRUE, False, "whether to normalize the images", {}),
    # fusion
    ("fuse_method", str, "cat", "fusion method of encoded IMU and Images [cat, soft, hard]", {}),
    # ODE
    ("ode_hidden_dim", int, 512, "size of the ODE latent", {}),
    ("ode_fn_num_layers", int, 3, "number of layers for the ODE", {}),
    ("ode_activation_fn", str, "tanh", "activation function [softplus, relu, leaky_relu, tanh]", {}),
    ("ode_solver", str, "dopri5", "ODE solvers [dopri5, heun, euler, tsit5, rk4 (=runge_kutta), rk4_classic]", {}),
    # RNN
    ("ode_rnn_type", str, "rnn", "type of RNN [rnn, gru]", {}),
    ("rnn_num_layers", int, 2, "number of layers for RNN", {}),
    ("rnn_hidden_dim", int, 1024, "size of the RNN latent (unused by the reference, PoseODERNN.py:43)", {}),
    ("rnn_dropout_out", float, 0, "dropout for the RNN output layer", {}),
    # CDE
    ("cde_hidden_dim", int, 128, "size of the CDE latent", {}),
    ("cde_fn_num_layers", int, 3, "number of layers for the CDE Function", {}),
    ("cde_num_layers", int, 3, "number of layers for the CDE", {}),
    ("cde_activation_fn", str, "tanh", "activation function [softplus, relu, leaky_relu, tanh]", {}),
    ("cde_solver", str, "dopri5", "ODE solvers [dopri5, heun, euler, rk4, tsit5]", {}),
    ("adjoint", _STORE_TRUE, False, "whether to use adjoint method", {}),
    # --- build extensions (not in the reference) ---
    ("ode_substeps", int, 1, "[ext] fixed sub-steps per observation interval (rk4/rk4_classic)", {}),
    ("dtype", str, "fp32", "[ext] arithmetic of the image encoder [fp32 (two fp16 pieces per operand, fp32-grade), fp32_mfma, fp16, bf16 (= fp16: reduced precision)]", {}),
]
# fmt: on


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for name, typ, default, helptext, extra in _FLAGS:
        if typ == _STORE_TRUE:
            parser.add_argument("--" + name, default=default, action="store_true", help=helptext)
        else:
            parser.add_argument("--" + name, type=typ, default=default, help=helptext, **extra)
    return parser


def get_args(argv=None):
    """Same call as the reference's ``get_args()`` (config.py:5); ``argv=None`` reads ``sys.argv``."""
    return build_parser().parse_args(argv)


def default_opt(**overrides):
    """Namespace with every default, then ``overrides`` applied (convenience for tests/bench)."""
    opt = build_parser().parse_args([])
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError(f"unknown option {k!r}")
        setattr(opt, k, v)
    return opt
