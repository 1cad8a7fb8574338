"""Checkpoint wire-compatibility with the reference.

``save_checkpoint`` mirrors src/train_utils.py:13-30 (same signature and file naming: the path is built by string
concatenation ``output_folder + filename + '_%03d.pth.tar'``, previous epoch deleted, ``_best`` copy), and
``load_checkpoint`` restores the dict written by experiments/rb2d/train.py:390-397
(keys epoch / unet_state_dict / imnet_state_dict / optim_state_dict / tracked_stats / global_step) into this
package's modules -- whose state_dict keys are identical to the reference's, including the duplicated ``fc.N.*``
entries of ImNet -- so reference-trained weights run on the HIP path unchanged.
"""
import os
import shutil

import torch


def save_checkpoint(state, is_best, epoch, output_folder, filename, logger=None):
    if epoch > 1:
        prev = output_folder + filename + '_%03d' % (epoch - 1) + '.pth.tar'
        if os.path.exists(prev):
            os.remove(prev)
    path = output_folder + filename + '_%03d' % epoch + '.pth.tar'
    torch.save(state, path)
    if is_best:
        if logger is not None:
            logger.info("Saving new best model")
        shutil.copyfile(path, output_folder + filename + '_best.pth.tar')
    return path


def _strip_module(sd):
    """Accept state dicts saved from nn.DataParallel / DDP wrappers as well."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_checkpoint(path, unet=None, imnet=None, optimizer=None, map_location="cpu"):
    """Load a reference-format checkpoint; returns the remaining bookkeeping (epoch, global_step, tracked_stats)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    if unet is not None:
        unet.load_state_dict(_strip_module(ck["unet_state_dict"]))
    if imnet is not None:
        imnet.load_state_dict(_strip_module(ck["imnet_state_dict"]))
    if optimizer is not None and "optim_state_dict" in ck:
        optimizer.load_state_dict(ck["optim_state_dict"])
        for st in optimizer.state.values():      # train.py:346-349: move optimizer state to the parameters' device
            for k, v in st.items():
                if torch.is_tensor(v) and k != "step":
                    p_dev = next(iter(optimizer.param_groups[0]["params"])).device
                    st[k] = v.to(p_dev)
    return {k: ck.get(k) for k in ("epoch", "global_step", "tracked_stats")}
