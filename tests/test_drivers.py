"""Drivers: config surface, log format, checkpoint keys.  CPU: BASELINE config 1 (MNIST ST, --no-cuda) against the
oracle; GPU: BASELINE configs 2-4 run for a couple of synthetic batches through the HIP path."""
import os
import re
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "edge-enhancement_amd")


def run_driver(ds_dir, script, cfg, out, *extra):
    cmd = [sys.executable, script, "-c", cfg, "--output-root", str(out), "--max-epochs", "1"] + list(extra)
    r = subprocess.run(cmd, cwd=os.path.join(PKG, ds_dir), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def read_log(out):
    logs = [os.path.join(d, f) for d, _, fs in os.walk(str(out)) for f in fs if f == "log.txt"]
    assert len(logs) == 1
    return logs[0], open(logs[0]).read().splitlines()


def parse_like_read_log(line):
    """utils/read_log.py:27-49 splits the summary lines on single spaces and takes fields 4 and 6."""
    parts = line.split(" ")
    return float(parts[4]), float(parts[6])


def test_every_config_has_the_reference_keys():
    need = {"method_name", "arch", "start_epoch", "epochs", "batch_size", "lr", "momentum", "weight_decay", "workers", "pin_memory",
            "print_freq", "seed", "epsilon", "num_steps_1", "step_size_1", "num_steps_2", "step_size_2", "num_steps_3", "step_size_3",
            "random", "alpha", "sigma", "w", "r", "gf", "low", "high", "type_canny", "n_queries"}
    n = 0
    for ds in ("MNIST/configs_mnist", "Tiny_ImageNet/configs_tinyimagenet", "ImageNet/configs_imagenet"):
        for f in sorted(os.listdir(os.path.join(PKG, ds))):
            text = open(os.path.join(PKG, ds, f)).read()
            d = yaml.safe_load(text)
            assert need <= set(d), (f, need - set(d))
            keys = re.findall(r"^(\w+):", text, flags=re.M)
            assert len(keys) == len(set(keys)), "duplicate key in " + f  # the reference's duplicate step_size_1 is resolved
            n += 1
    assert n == 31
    t = yaml.safe_load(open(os.path.join(PKG, "Tiny_ImageNet/configs_tinyimagenet/trades_training.yml")))
    assert abs(t["step_size_1"] - 1 / 255) < 1e-9 and t["beta"] == 6.0  # effective value
    e = yaml.safe_load(open(os.path.join(PKG, "Tiny_ImageNet/configs_tinyimagenet/ee_at_bpda3_square.yml")))
    assert abs(e["step_size_1"] - 2 / 255) < 1e-9 and e["type_canny"] == "CannyFilter_step125_1" and e["r"] == 8 and e["high"] == 76.0


def test_mnist_standard_training_on_cpu_matches_oracle(tmp_path):
    """BASELINE config 1: MNIST LeNet ST via experiments_mnist.py --no-cuda (plumbing path, no GPU)."""
    stdout = run_driver("MNIST", "experiments_mnist.py", "configs_mnist/standard_training.yml", tmp_path, "--no-cuda", "--data", "synthetic:2:1")
    path, lines = read_log(tmp_path)
    assert "/checkpoint_MNIST/ST/Net2/None-bs50-lr0.1-momentum0.3-wd0.0001-seed1/log/log.txt" in path
    m = re.match(r"Epoch: \[0\]\[0/2\]\tTime [\d.]+ \([\d.]+\)\tData [\d.]+ \([\d.]+\)\tLoss ([\d.]+) \(([\d.]+)\)\tPrec@1 ([\d.]+) \(([\d.]+)\)\t"
                 r"Prec@5 ([\d.]+) \(([\d.]+)\)\t$", lines[0])
    assert m, lines[0]
    assert m.group(3) == m.group(5)  # MNIST driver logs prec1 as top5 (experiments_mnist.py:246)
    clean = [l for l in lines if l.startswith(" * Clean")]
    adv = [l for l in lines if l.startswith(" * Adv")]
    assert len(clean) == 1 and len(adv) == 1
    c1, c5 = parse_like_read_log(clean[0])
    a1, a5 = parse_like_read_log(adv[0])
    assert 0 <= a1 <= c1 <= 100 and a5 <= c5 <= 100
    assert any(l.startswith("Test_clean: [0/1]\tTime") for l in lines) and any(l.startswith("Test_adv: [0/1]\tTime") for l in lines)
    # first training loss == the oracle's on the same seeded model and batch
    from oracle import ref_path as R
    from utils.helper import set_seed
    from eeadv.driver import SyntheticLoader
    set_seed(1)
    net = R.Net_2().train()
    x, y = SyntheticLoader(2, 50, (1, 28, 28), 10, "cpu", 1001).batches[0]
    want = F.cross_entropy(net(x), y).item()
    assert abs(float(m.group(1)) - want) < 5e-5
    ck = [os.path.join(d, f) for d, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith(".pth")]
    assert any(os.path.basename(c) == "at_numstep40_epsilon76_r0_canny_sigma0_alpha0-bs50-lr_0.1-w0-gfFalse-l0-h0_0.pth" for c in ck)
    state = torch.load(ck[0], weights_only=True)
    assert set(state) == {"epoch", "arch", "state_dict", "best_prec1", "optimizer"} and state["epoch"] == 1 and state["arch"] == "Net2"
    # nn.DataParallel(model).state_dict() in the reference (experiments_mnist.py:77,168): `module.`-prefixed keys, so that the
    # reference's own --resume (which loads into the wrapped model) accepts the file; ours strips the prefix again
    assert "module.conv1.weight" in state["state_dict"] and "conv1.weight" not in state["state_dict"]
    first = [c for c in ck if "/model_pth/" in c][0]
    r = subprocess.run([sys.executable, "experiments_mnist.py", "-c", "configs_mnist/standard_training.yml", "--no-cuda", "--data", "synthetic:2:1",
                        "--output-root", str(tmp_path / "resumed"), "--resume", first, "--max-epochs", "1"],
                       cwd=os.path.join(PKG, "MNIST"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "=> loaded checkpoint" in r.stdout and "(epoch 1)" in r.stdout and "Epoch: [1][0/2]" in r.stdout


def test_unknown_arch_and_real_data_are_refused(tmp_path):
    cfg = tmp_path / "bad.yml"
    base = open(os.path.join(PKG, "MNIST/configs_mnist/standard_training.yml")).read()
    cfg.write_text(base.replace("arch: 'Net2'", "arch: 'Net9'"))
    r = subprocess.run([sys.executable, "experiments_mnist.py", "-c", str(cfg), "--no-cuda", "--output-root", str(tmp_path)],
                       cwd=os.path.join(PKG, "MNIST"), capture_output=True, text=True)
    assert r.returncode != 0 and "NotImplementedError" in r.stderr
    r = subprocess.run([sys.executable, "experiments_mnist.py", "-c", "configs_mnist/standard_training.yml", "--no-cuda", "--data", "/data/mnist",
                        "--output-root", str(tmp_path)], cwd=os.path.join(PKG, "MNIST"), capture_output=True, text=True)
    assert r.returncode != 0 and "synthetic" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ds,script,cfg", [
    ("MNIST", "experiments_mnist.py", "configs_mnist/ee_at_bpda3_square.yml"),               # BASELINE config 2
    ("Tiny_ImageNet", "experiments_tinyimagenet.py", "configs_tinyimagenet/trades_training.yml"),      # BASELINE config 3
    ("Tiny_ImageNet", "experiments_tinyimagenet.py", "configs_tinyimagenet/ee_at_bpda3_square.yml"),   # BASELINE config 4 (1 rank)
    ("Tiny_ImageNet", "experiments_tinyimagenet.py", "configs_tinyimagenet/avmixup_training.yml"),
    ("Tiny_ImageNet", "experiments_tinyimagenet.py", "configs_tinyimagenet/alp_training.yml"),
    ("Tiny_ImageNet", "experiments_tinyimagenet.py", "configs_tinyimagenet/targeted_ee_at_bpda3_square.yml"),
])
def test_gpu_configs_run_through_the_hip_path(tmp_path, ds, script, cfg):
    run_driver(ds, script, cfg, tmp_path, "--data", "synthetic:2:1")
    _, lines = read_log(tmp_path)
    assert lines[0].startswith("Epoch: [0][0/2]\tTime")
    loss = float(re.search(r"Loss ([\d.]+) ", lines[0]).group(1))
    assert 0.5 < loss < 40
    c1, _ = parse_like_read_log([l for l in lines if l.startswith(" * Clean")][0])
    a1, _ = parse_like_read_log([l for l in lines if l.startswith(" * Adv")][0])
    assert 0 <= a1 <= 100 and 0 <= c1 <= 100


def test_free_at_scripts_keep_the_reference_command_line(tmp_path):
    """Both free-AT scripts on CPU (no launch): every flag of the reference's parsers (AT_free_imagenet_ddp.py:38-108,
    AT_hfs_canny_free_imagenet_ddp.py:39-115) with its default, directory layout and checkpoint names (:172-173 / :197, :241-254)."""
    import importlib
    sys.path.insert(0, os.path.join(PKG, "ImageNet", "free_imagenet"))
    try:
        base = importlib.import_module("AT_free_imagenet_ddp")
        ee = importlib.import_module("AT_hfs_canny_free_imagenet_ddp")
    finally:
        sys.path.pop(0)
    a = base.make_parser().parse_args([])
    assert (a.arch, a.epochs, a.batch_size, a.lr, a.momentum, a.weight_decay, a.print_freq) == ("resnet152", 90, 256, 0.1, 0.9, 1e-4, 100)
    assert (a.epsilon, a.num_steps_1, a.step_size_1, a.clip_eps, a.fgsm_step, a.n_repeats, a.crop_size) == (4 / 255, 50, 1 / 255, 4.0, 4.0, 4, 224)
    assert (a.w, a.r, a.low, a.high, a.sigma, a.alpha, a.gf, a.seed, a.max_color_value) == (0, 0, 0, 0, 0, 0, False, 1, 255.0)
    e = ee.make_parser().parse_args([])
    assert (e.arch, e.num_steps_1, e.num_steps_2, e.num_steps_3, e.step_size_3) == ("resnet50_EE_square", 10, 50, 100, 1 / 255)
    assert (e.w, e.r, e.low, e.high, e.sigma, e.type_canny, e.n_queries) == (1, 16, 38, 76, 1, "CannyFilter_step125_1", 1)
    with pytest.raises(SystemExit):
        base.make_parser().parse_args(["-a", "resnet50_EE_square"])  # the plain script does not know the EE models
    for args, mod in ((a, base), (e, ee)):
        args.output_root, args.clip_eps, args.fgsm_step = str(tmp_path), args.clip_eps / 255, args.fgsm_step / 255
    d = base.output_dirs(a)
    assert d["model"].endswith("/checkpoint_free_imagenet/free_AT_ddp/resnet152Baseline_clip-eps4/model_pth/")
    d2 = ee.output_dirs(e)
    assert d2["log"].endswith("/checkpoint_free_imagenet/free_AT_ddp/resnet50_EE_square/CannyFilter_step125_1_clip-eps4/log/") and os.path.isdir(d2["log"])
    f, best = base.checkpoint_names(e, d2, 7)
    assert f.endswith("model_pth/at_clip-eps4_fgsm-step4_n-repeats4_r16_canny_sigma1_alpha0-bs256-lr_0.1-w1-gfFalse-l38-h76-ty1_7.pth")
    assert best.endswith("best_model_pth/at_clip-eps4_fgsm-step4_n-repeats4_r16_canny_sigma1_alpha0-bs256-lr_0.1-w1-gfFalse-l38-h76-ty1_.pth")


@pytest.mark.gpu
def test_free_at_script_runs_config5_on_one_rank(tmp_path):
    """BASELINE config 5 (ImageNet/free_imagenet/AT_free_imagenet_ddp.py) on one rank: resnet50, 224x224, per-rank batch 32,
    2 batches x 4 repeats through the HIP kernels, the PGD evaluation (shortened to 2 steps), checkpoint with the
    reference's keys / `module.`-prefixed names / file name, then --resume and --evaluate from it."""
    script = os.path.join(PKG, "ImageNet", "free_imagenet", "AT_free_imagenet_ddp.py")
    base = [sys.executable, script, "-a", "resnet50", "-b", "32", "--data", "synthetic:2:1", "--print-freq", "1", "--num-steps-1", "2",
            "--output-root", str(tmp_path)]
    r = subprocess.run(base + ["--max-epochs", "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout
    assert "clip-eps:4,fgsm-step:4,n-repeats:4" in out and "epochs:23" in out  # ceil(90 / 4), :129
    ep = [l for l in out.splitlines() if l.startswith("Epoch: [0]")]
    assert len(ep) == 2 and ep[1].startswith("Epoch: [0][1/2]\tTime")
    loss = float(re.search(r"Loss ([\d.]+) ", ep[0]).group(1))
    assert 5.0 < loss < 12.0  # ln(1000) = 6.9 at initialisation
    assert any(l.startswith(" * Adv Prec@1") for l in out.splitlines()) and any(l.startswith(" * Ad Prec@1") for l in out.splitlines())
    ck = [os.path.join(d, f) for d, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith(".pth")]
    name = "at_clip-eps4_fgsm-step4_n-repeats4_r0_canny_sigma0_alpha0-bs32-lr_0.1-w0-gfFalse-l0-h0-ty1_0.pth"
    first = [c for c in ck if c.endswith("/resnet50Baseline_clip-eps4/model_pth/" + name)]
    assert first, ck
    state = torch.load(first[0], weights_only=True)
    assert set(state) == {"epoch", "arch", "state_dict", "best_prec1", "optimizer"} and state["epoch"] == 1 and state["arch"] == "resnet50"
    assert "module.layer4.2.bn3.running_var" in state["state_dict"] and "module.fc.weight" in state["state_dict"]
    r = subprocess.run(base + ["--resume", first[0], "--max-epochs", "0", "--evaluate"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "=> loaded checkpoint" in r.stdout and " * Clean Prec@1" in r.stdout and "Epoch: [" not in r.stdout


@pytest.mark.gpu
def test_free_at_ee_script_runs_on_one_rank(tmp_path):
    """ImageNet/free_imagenet/AT_hfs_canny_free_imagenet_ddp.py on one rank at its defaults: `-a
    resnet50_EE_square` (which, as in the reference, builds resnet18_EE_square), 224 x 224, HighFreqSuppress r = 16 on the band
    kernel, CannyFilter_step125_1, 2 batches x 4 repeats, evaluation with the num_steps_3 attack (shortened), the
    `<arch>/<type_canny>_clip-eps<e>/` layout, the argument dump at the head of log.txt, the checkpoint name with the integer
    defaults of the reference (`sigma1`, `w1`, `l38`, `h76`)."""
    script = os.path.join(PKG, "ImageNet", "free_imagenet", "AT_hfs_canny_free_imagenet_ddp.py")
    cmd = [sys.executable, script, "-b", "16", "--data", "synthetic:2:1", "--print-freq", "1", "--num-steps-3", "2", "--num-classes", "100",
           "--output-root", str(tmp_path), "--max-epochs", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout
    assert "=> creating model 'resnet50_EE_square'" in out and "CannyFilter; sigma:1" in out
    ep = [l for l in out.splitlines() if l.startswith("Epoch: [0]")]
    assert len(ep) == 2
    loss = float(re.search(r"Loss ([\d.]+) ", ep[0]).group(1))
    assert 0.5 < loss < 9.0  # ln(100) = 4.6 at initialisation; the first line is printed after the batch's four updates
    assert any(l.startswith(" * Adv Prec@1") for l in out.splitlines())
    root = os.path.join(str(tmp_path), "checkpoint_free_imagenet", "free_AT_ddp", "resnet50_EE_square", "CannyFilter_step125_1_clip-eps4")
    name = "at_clip-eps4_fgsm-step4_n-repeats4_r16_canny_sigma1_alpha0-bs16-lr_0.1-w1-gfFalse-l38-h76-ty1_0.pth"
    assert os.path.isfile(os.path.join(root, "model_pth", name)), os.listdir(os.path.join(root, "model_pth"))
    log = open(os.path.join(root, "log", "log.txt")).read()
    assert log.startswith("Namespace(") and "type_canny='CannyFilter_step125_1'" in log.splitlines()[0] and "Epoch: [0][1/2]" in log
    state = torch.load(os.path.join(root, "model_pth", name), weights_only=True)
    assert state["arch"] == "resnet50_EE_square" and "module.layer4.1.bn2.running_var" in state["state_dict"]  # two blocks: resnet18
    assert "module.layer4.2.bn1.weight" not in state["state_dict"]


@pytest.mark.gpu
@pytest.mark.parametrize("grad_sync", ["flat", "ddp"])
def test_free_at_script_two_ranks_share_one_gpu(tmp_path, grad_sync):
    """`main()` of the free-AT script itself at world size 2 - the branch that converts to SyncBatchNorm and builds the gradient
    exchange, which the one-rank test never enters.  Two ranks time-share cuda:0 and exchange over gloo (EEADV_SHARE_GPU /
    EEADV_DIST_BACKEND); small shapes (resnet18, 224 x 224, 10 classes, global batch 8), one batch x 4 repeats, the PGD evaluation, the
    checkpoint; both gradient-exchange modes."""
    script = os.path.join(PKG, "ImageNet", "free_imagenet", "AT_free_imagenet_ddp.py")
    env = dict(os.environ, EEADV_SHARE_GPU="1", EEADV_DIST_BACKEND="gloo", EEADV_GRAD_SYNC=grad_sync, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300) + (0 if grad_sync == "flat" else 301)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           script, "-a", "resnet18", "-b", "8", "--num-classes", "10", "--data", "synthetic:1:1", "--print-freq", "1",
           "--num-steps-1", "1", "--max-epochs", "1", "--output-root", str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = r.stdout
    assert "n-repeats:4,world:2" in out
    ep = [l for l in out.splitlines() if l.startswith("Epoch: [0]")]
    assert len(ep) == 1  # rank 0 prints
    loss = float(re.search(r"Loss ([\d.]+) ", ep[0]).group(1))
    assert 0.5 < loss < 20.0  # ln(10) = 2.3 at initialisation; the line is printed after the batch's four updates at lr 0.1 on random labels
    assert any(l.startswith(" * Adv Prec@1") for l in out.splitlines())
    ck = [os.path.join(d, f) for d, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith("_0.pth")]
    assert len(ck) == 1, ck
    state = torch.load(ck[0], weights_only=True)
    assert "module.layer4.1.bn2.running_var" in state["state_dict"] and state["epoch"] == 1
