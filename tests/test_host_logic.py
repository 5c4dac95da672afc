"""Host-side logic on CPU: flag surface, state_dict keys, error behaviour, synthetic inputs."""
import os
import sys

import pytest
import torch

from odevio_amd import config, default_opt, synth, weights

REF = "/root/reference"


def test_defaults_match_the_reference_hot_path_flags():
    o = default_opt()
    assert (o.model_type, o.img_w, o.img_h, o.v_f_len, o.i_f_len, o.seq_len) == ("ode-rnn", 512, 256, 512, 256, 11)
    assert (o.fuse_method, o.ode_hidden_dim, o.ode_fn_num_layers, o.ode_activation_fn, o.ode_solver) == \
        ("cat", 512, 3, "tanh", "dopri5")
    assert (o.ode_rnn_type, o.rnn_num_layers, o.rnn_hidden_dim, o.batch_size) == ("rnn", 2, 1024, 26)
    assert (o.cde_hidden_dim, o.cde_fn_num_layers, o.cde_solver, o.adjoint) == (128, 3, "dopri5", False)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_flag_surface_equals_the_reference_parser():
    sys.path.insert(0, REF)
    try:
        import importlib
        ref_cfg = importlib.import_module("scripts.config")
        argv, sys.argv = sys.argv, ["x"]
        try:
            ref = vars(ref_cfg.get_args())
        finally:
            sys.argv = argv
    finally:
        sys.path.remove(REF)
    ours = vars(config.get_args([]))
    for k, v in ref.items():
        assert k in ours, k
        if k in ("ode_solver",):
            continue
        assert ours[k] == v, (k, ours[k], v)
    assert set(ours) - set(ref) == {"ode_substeps", "dtype"}


def test_reference_command_line_parses():
    # the reference's own training recipe (scripts/run_training.sh:6-28)
    o = config.get_args("--ode_activation_fn=softplus --ode_fn_num_layers=2 --ode_solver=dopri5 --rnn_num_layers=3 "
                        "--ode_hidden_dim=1024 --fuse_method=soft --freeze_encoder --data_dropout=0.3 "
                        "--train_seq 04 --val_seq 04".split())
    assert o.ode_hidden_dim == 1024 and o.rnn_num_layers == 3 and o.freeze_encoder and o.train_seq == ["04"]


def test_state_dict_keys_and_shapes():
    sd = weights.make_state_dict(default_opt(fuse_method="soft"))
    assert sd["Image_net.conv1.0.weight"].shape == (64, 6, 7, 7)
    assert sd["Image_net.conv6.1.running_var"].shape == (1024,) and abs(float(sd["Image_net.conv6.1.running_var"][0]) - 0.9) < 1e-6
    assert sd["Image_net.visual_head.weight"].shape == (512, 32768)
    assert sd["Inertial_net.encoder_conv.8.weight"].shape == (256, 128, 3)
    assert sd["Inertial_net.proj.weight"].shape == (256, 2816)
    assert [k for k in sd if k.startswith("Pose_net.ode_func")] == [
        f"Pose_net.ode_func.net.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    assert sd["Pose_net.ode_func.net.0.weight"].shape == (512, 768) and sd["Pose_net.ode_func.net.6.weight"].shape == (768, 512)
    assert sd["Pose_net.rnn.weight_hh_l1"].shape == (768, 768) and sd["Pose_net.fuse.net.0.weight"].shape == (768, 768)
    assert sd["Pose_net.regressor.2.weight"].shape == (6, 128)
    gru = weights.make_state_dict(default_opt(ode_rnn_type="gru"))
    assert gru["Pose_net.rnn.weight_ih_l0"].shape == (2304, 768)
    # seeded and order-independent
    again = weights.make_state_dict(default_opt(fuse_method="soft"))
    assert all(torch.equal(sd[k], again[k]) for k in sd)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_state_dict_loads_strictly_into_reference_modules():
    sys.path.insert(0, REF)
    try:
        from src.models.Encoder import ImageEncoder, InertialEncoder
        from src.models.PoseRNN import PoseRNN
    finally:
        sys.path.remove(REF)
    opt = default_opt(img_h=64, img_w=128, model_type="rnn", fuse_method="soft", ode_rnn_type="gru", rnn_num_layers=3)
    sd = weights.make_state_dict(opt, seed=3)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    ImageEncoder(opt).load_state_dict(sub("Image_net."), strict=True)
    InertialEncoder(opt).load_state_dict(sub("Inertial_net."), strict=True)
    PoseRNN(opt).load_state_dict(sub("Pose_net."), strict=True)


def test_model_container_has_reference_surface_and_checkpoint_aliases():
    from odevio_amd import DeepVIO
    opt = default_opt(img_h=64, img_w=128)
    m = DeepVIO(opt)
    assert all(hasattr(m, a) for a in ("Image_net", "Inertial_net", "Pose_net", "opt"))
    assert len(list(m.Pose_net.get_regressor_params())) == 4
    assert all(not n.startswith("regressor") for n, _ in m.Pose_net.named_parameters()
               if any(p is q for q in m.Pose_net.get_other_params() for p in [dict(m.Pose_net.named_parameters())[n]]))
    sd = dict(m.state_dict())
    # a reference checkpoint repeats ODEFunc under the torch.compile'd solver and may carry a DataParallel prefix
    ckpt = {"module." + k: v for k, v in sd.items()}
    ckpt["module.Pose_net.solver._orig_mod.step_method.term.f.net.0.weight"] = sd["Pose_net.ode_func.net.0.weight"]
    m.load_state_dict(ckpt, strict=True)


def test_errors_follow_the_reference():
    from odevio_amd import DeepVIO
    for bad in (dict(ode_solver="rk45"), dict(ode_rnn_type="lstm"), dict(ode_activation_fn="gelu"), dict(model_type="foo")):
        with pytest.raises(ValueError):
            DeepVIO(default_opt(**bad))
    with pytest.raises(NotImplementedError):
        DeepVIO(default_opt(model_type="ltc"))
    m = DeepVIO(default_opt(img_h=64, img_w=128))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 2, 3, 64, 128), torch.zeros(1, 11, 6), torch.zeros(1, 2))


def test_synthetic_timestamps():
    reg = synth.timestamps(4, 11)
    assert torch.allclose(reg, 0.1 * torch.arange(11).float().expand(4, 11))
    irr = synth.timestamps(64, 11, drop=0.5, seed=1, absolute=True)
    d = irr.double().diff(dim=1)
    assert (d > 0.05).all() and d.max() > 0.25  # strictly ascending, multiples of the frame period with gaps
    assert ((d / 0.1).round() - d / 0.1).abs().max() < 1e-2
    assert synth.imu(2, 11).shape == (2, 101, 6) and synth.images(1, 3, 64, 128).abs().max() <= 0.5


def test_flownet_checkpoint_loads_by_key_intersection():
    """scripts/train_model.py:180-188: a FlowNet checkpoint ({"state_dict": ...}) carries layers the encoder does not own
    (decoder, flow prediction) and lacks the encoder's visual_head; only the common keys are taken, the rest of Image_net
    stays as it was."""
    import torch
    from odevio_amd import DeepVIO, default_opt, weights
    opt = default_opt(img_h=64, img_w=128)
    model = DeepVIO(opt, seed=3)
    before = {k: v.clone() for k, v in model.Image_net.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    flownet = {k: torch.randn(v.shape, generator=g) for k, v in before.items() if k.startswith(("conv1.", "conv2.", "conv3.", "conv6."))
               and v.is_floating_point()}
    flownet["deconv5.0.weight"] = torch.randn(1024, 512, 4, 4, generator=g)     # FlowNetS decoder: not ours
    flownet["predict_flow6.weight"] = torch.randn(2, 1024, 3, 3, generator=g)
    taken = weights.load_flownet_checkpoint(model, {"state_dict": flownet, "epoch": 12})
    after = model.Image_net.state_dict()
    assert taken == sorted(k for k in flownet if k in before) and len(taken) > 10
    for k in before:
        if k in flownet:
            assert torch.equal(after[k], flownet[k]), k
        else:
            assert torch.equal(after[k], before[k]), k        # conv3_1, conv4.., visual_head untouched
    assert model._plan_sig is None
    # a wrong shape must fail as it does in the reference (load_state_dict)
    bad = {"state_dict": {"conv1.0.weight": torch.zeros(64, 6, 3, 3)}}
    with pytest.raises(RuntimeError):
        weights.load_flownet_checkpoint(model, bad)


def test_trainer_schedule_and_parameter_groups_follow_the_reference():
    """PoseNetTrainer.set_epoch = update_status (scripts/train_model.py:25-35) applied to parameter group 0 only (:214-215);
    needs no GPU: the schedule is host logic."""
    from odevio_amd import default_opt, train

    class _Model:
        opt = default_opt()

    t = train.PoseNetTrainer.__new__(train.PoseNetTrainer)
    t.model = _Model()
    t.lr = t.lr_regressor = 1e-4
    o = _Model.opt
    assert t.set_epoch(0) == o.lr_warmup and t.set_epoch(o.epochs_warmup - 1) == o.lr_warmup
    assert t.set_epoch(o.epochs_warmup) == o.lr_joint
    assert t.set_epoch(o.epochs_warmup + o.epochs_joint - 1) == o.lr_joint
    assert t.set_epoch(o.epochs_warmup + o.epochs_joint) == o.lr_fine
    assert t.lr_regressor == 1e-4          # the reference never re-assigns group 1
    names = train.fuse_param_names(default_opt(fuse_method="soft")) + train.pose_param_names(default_opt(fuse_method="soft"))
    assert sum(n.startswith("Pose_net.regressor.") for n in names) == 4 and "Pose_net.fuse.net.0.weight" in names
