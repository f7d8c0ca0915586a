"""The build's own counterpart of core/epoch_loops.py:8-83 (the reference file itself stays
untouched and runs on this package's FullModel / ChamferLoss as-is — see INTEGRATION.md).  Same
signatures and return values; differences: the loss bookkeeping accumulates properly instead of
doubling the last batch (epoch_loops.py:32-36), and no tqdm.
"""
import numpy as np
import torch

from ..model.full_model import FullModel


def train_epoch(epoch, full_model: FullModel, optimizer, loader, device, rec_loss_function, loss_coef=0.05):
    full_model.train()
    sum_all = sum_r = sum_kld = 0.0
    i = 0
    existing = gt = reconstruction = None
    for i, point_data in enumerate(loader, 1):
        optimizer.zero_grad()

        existing, missing, gt, _ = point_data

        existing = existing.to(device)
        missing = missing.to(device)
        gt = gt.to(device)

        reconstruction, logvar, mu = full_model(existing, missing, list(gt.shape), epoch, device)

        loss_r = torch.mean(
            loss_coef * rec_loss_function(gt, reconstruction.permute(0, 2, 1)))

        if full_model.mode.has_generativity():
            loss_kld = 0.5 * (torch.exp(logvar) + torch.square(mu) - 1 - logvar).sum()
            loss_kld = torch.div(loss_kld, existing.shape[0])
            loss_all = loss_r + loss_kld
            sum_kld += loss_kld.item()
        else:
            loss_all = loss_r
        sum_r += loss_r.item()
        sum_all += loss_all.item()

        loss_all.backward()
        optimizer.step()

    n = max(i, 1)
    return full_model, optimizer, sum_all / n, sum_kld / n, sum_r / n, \
        existing.detach().cpu().numpy(), gt.detach().cpu().numpy(), reconstruction.detach().cpu().numpy()


def val_epoch(epoch, full_model, device, loaders_dict, val_classes_names, loss_function, loss_coef=0.05):
    full_model.eval()

    val_losses = dict.fromkeys(val_classes_names)
    val_samples = dict.fromkeys(val_classes_names)

    with torch.no_grad():
        for cat_name, dl in loaders_dict.items():
            loss = 0.0
            i = 1
            for i, point_data in enumerate(dl, 1):
                existing, missing, gt, _ = point_data
                existing = existing.to(device)
                missing = missing.to(device)
                gt = gt.to(device)

                reconstruction = full_model(existing, missing, list(gt.shape), epoch, device)

                loss_our_cd = torch.mean(
                    loss_coef * loss_function(gt, reconstruction.permute(0, 2, 1)))

                loss += loss_our_cd.item()

            existing = existing.detach().cpu().numpy()
            gt = gt.detach().cpu().numpy()
            reconstruction = reconstruction.detach().cpu().numpy()

            val_samples[cat_name] = (existing[0], gt[0], reconstruction[0])
            val_losses[cat_name] = np.array([loss / i])

        total = np.zeros(1)
        for v in val_losses.values():
            total = np.add(total, v)
        val_losses['total'] = total / len(val_losses.keys())

    return val_losses, val_samples
