"""The two-phase throughput driver (train_ddp.py) end to end on synthetic data: schedule of train.py
(encoder-only epochs, then the full loss), its log lines, its final checkpoint format, plus per-epoch checkpoints and resume."""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd')


def run(out, *extra):
    cmd = [sys.executable, os.path.join(PKG, 'train_ddp.py'), '--de_type', 'denoising_25', '--degradation_embedding_method', 'all_3_bands',
           '--contrast_loss_weight', '0.6', '--compute_dtype', 'bf16', '--per_gpu_batch', '2', '--synthetic_steps', '2',
           '--output_path', out, '--epochs_encoder', '1', '--test_de_type', 'denoising_bsd68_25', 'deraining', *extra]
    r = subprocess.run(cmd, cwd=PKG, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_two_phase_training_and_resume(tmp_path):
    out = str(tmp_path) + '/'
    run(out, '--epochs', '3', '--save_every', '2')
    log = open(out + 'train.log').read().splitlines()
    assert re.fullmatch(r'Epoch \(0\)  Loss: contrast_loss:\d+\.\d{4}', log[0])                       # train.py:98-106
    assert re.fullmatch(r'Epoch \(1\)  Loss: l1_loss:\d+\.\d{4} contrast_loss:\d+\.\d{4}', log[1])    # train.py:107-117
    assert len(log) == 3
    assert os.path.exists(out + 'options.log')
    res = open(out + 'results.log').read().splitlines()                                                # train.py:131-139
    assert res[0] == '2 Epochs Results:' and res[3] == '3 Epochs Results:' and len(res) == 6
    m = re.fullmatch(r'denoising_bsd68_25: {8}PSNR/SSIM: (\d+\.\d{2})/(\d\.\d{4})', res[1])           # val_utils.py:50-66: both metrics
    assert m and 5 < float(m.group(1)) < 60 and 0 < float(m.group(2)) < 1, res[1]
    assert res[2].startswith('deraining: ' + ' ' * 16 + 'PSNR/SSIM: ')
    for e in (2, 3):
        assert os.path.exists(out + f'ckpt/epoch_{e}.pth') and os.path.exists(out + f'ckpt/epoch_{e}.opt.pth')
    sd = torch.load(out + 'ckpt/epoch_3.pth', map_location='cpu', weights_only=True)
    assert len(sd) == 2496 and all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    st = torch.load(out + 'ckpt/epoch_2.opt.pth', map_location='cpu', weights_only=True)
    assert float(st['hyper'][3]) == 4.0 and float(st['hyper_rest'][3]) == 2.0      # Adam steps: encoder 2 epochs x 2, the rest 1 x 2
    # resume from epoch 2 and train epoch index 2 again: same number of log lines afterwards (appended), finite losses
    run(out, '--epochs', '3', '--resume', out + 'ckpt/epoch_2.pth')
    log2 = open(out + 'train.log').read().splitlines()
    assert len(log2) == 4 and log2[3].startswith('Epoch (2)  Loss: l1_loss:')


def test_frequency_l1_term(tmp_path):
    """--num_frequency_bands_l1 (train.py:69-70,90-91): the extra L1 on band spectra back-propagates through the differentiable
    FrequencyDecompose and shows up in the logged l1_loss."""
    out = str(tmp_path) + '/'
    a = run(out + 'a/', '--epochs', '1', '--epochs_encoder', '0')
    b = run(out + 'b/', '--epochs', '1', '--epochs_encoder', '0', '--num_frequency_bands_l1', '3')
    la = float(re.search(r'l1_loss:(\d+\.\d+)', open(out + 'a/train.log').read()).group(1))
    lb = float(re.search(r'l1_loss:(\d+\.\d+)', open(out + 'b/train.log').read()).group(1))
    assert lb > la > 0 and lb < 100 * la + 10, (la, lb)


def test_mixed_degradation_batches(tmp_path):
    """BASELINE configs[2]: `--de_type` with several tasks -- every sample of a batch carries the next task in turn
    (dataset_utils.py:99), rain / haze from the synthetic stand-ins of fwair/augment.py."""
    out = str(tmp_path) + '/'
    run(out, '--epochs', '1', '--epochs_encoder', '0', '--per_gpu_batch', '3', '--de_type', 'denoising_15', 'deraining', 'dehazing')
    line = open(out + 'train.log').read().splitlines()[0]
    m = re.fullmatch(r'Epoch \(0\)  Loss: l1_loss:(\d+\.\d{4}) contrast_loss:(\d+\.\d{4})', line)
    assert m and 0 < float(m.group(1)) < 1 and 0 < float(m.group(2)) < 20
