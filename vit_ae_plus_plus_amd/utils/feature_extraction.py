"""Feature dump after pre-training (reference: utils/feature_extraction.py:9-45): run the encoder-only model over a
loader and save the stacked features / labels as .npy.  Same signature and files; the features come from
``VisionTransformer3D.forward_features`` on the HIP kernels (the reference's autocast region corresponds to
``model.set_precision('bf16')``)."""
import os

import numpy as np
import torch


@torch.no_grad()
def generate_features(data_loader, model, device, ssl_feature_dir, feature_file_name='features.npy',
                      label_file_name='gt_labels.npy', log_writer=None):
    model.eval()
    feats, labels = [], []
    for batch in data_loader:
        images, target = batch[0], batch[-1]
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        feats.append(model.forward_features(images))
        labels.append(target)
    out_pred = torch.cat(feats, 0) if feats else torch.empty(0, device=device)
    out_gt = torch.cat(labels, 0).float() if labels else torch.empty(0, device=device)
    if feature_file_name is not None:
        print("Saving features!!!")
        np.save(os.path.join(ssl_feature_dir, feature_file_name), out_pred.cpu().numpy())
    if label_file_name is not None:
        print("Saving labels!!!")
        np.save(os.path.join(ssl_feature_dir, label_file_name), out_gt.cpu().numpy())
    if log_writer is not None:
        metadata = [x.item() for x in out_gt]
        log_writer.add_embedding(out_pred, metadata=metadata, tag='ssl_embedding')
