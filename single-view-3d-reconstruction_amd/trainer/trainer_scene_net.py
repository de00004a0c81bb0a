"""Mirror of the reference's SceneNetTrainer forward / training_step / losses contract
(trainer/trainer_scene_net.py:22-55,69-119,145-168) -- BASELINE config 5 -- without Lightning:

    rgb -> Unet -> resize 320 / crop rows 40:280 -> sigmoid*(max_z-min_z)+min_z          (:71-80)
        -> project.depthmap_to_gridspace -> norm_grid_space -> project() voxel occupancy (:85-88)
        -> IFNet(voxel_occupancy, points)                                                 (:101)
    loss = BCE(mean) + MSE(depth, depthmap_target)   (or BCE only with no_depth_sup)      (:147-168)
    Adam groups: unet lr, project 10*lr, ifnet lr                                        (:45-55)

The UNet is stock PyTorch-ROCm ops; unprojection, splat, blur, encoder, gather,
MLP and the BCE run in the HIP kernels.  `subsample_points != 0` (:91-99,108-114): the projected point cloud is
queried too and labelled against the sample's mesh ON THE DEVICE (..data_processing.mesh_occupancies.determine_occupancy,
SURVEY §8 f3) -- the reference copies it to the host and runs trimesh + Cython + numpy per step.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..model.ifnet import IFNet
from ..model.projection import project
from ..model.unet import UNetMini, Unet
from ..data_processing.mesh_occupancies import determine_occupancy


class _BCELogitsMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        B, N = logits.shape
        loss, dz = ops.bce_logits_sum_mean(logits.contiguous(), targets.contiguous().float(), want_grad=True,
                                           gscale=1.0 / N)
        ctx.save_for_backward(dz)
        return loss.squeeze(0) / N

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None


def default_hparams(**kw):
    h = dict(lr=1e-4, kernel_size=[3, 3, 3], sigma=[1.5, 1.5, 1.5], scale_factor=1, resize_input=True, skip_unet=False,
             subsample_points=0, no_depth_sup=False, min_z=0.1953997164964676, max_z=7.0, net_res=128,
             reference_occupancy_quirk=True, miopen_benchmark=True)
    h.update(kw)
    return SimpleNamespace(**h)


class SceneNetTrainer(nn.Module):
    def __init__(self, kwargs=None, dims=None):
        super().__init__()
        self.hparams = kwargs if kwargs is not None else default_hparams()
        h = self.hparams
        self.ifnet = IFNet(net_res=getattr(h, "net_res", 128))
        self.kernel_size = h.kernel_size
        if dims is None:
            dims = (torch.tensor([139, 104, 112]) / h.scale_factor).round().long()
        self.dims = torch.as_tensor(dims).long()
        self.project = project(self.dims, self.kernel_size, torch.tensor(h.sigma, dtype=torch.float32))
        if not h.skip_unet and getattr(h, "miopen_benchmark", True):
            # The UNet's stock MIOpen convolutions fall back to `naive_conv_*` solvers on gfx950 unless MIOpen is allowed
            # to time its solvers once per shape: config-5 step 79 -> 27 ms (tools/bench_scene.py).  Process-wide switch.
            torch.backends.cudnn.benchmark = True
        if h.skip_unet:
            self.unet = None
        elif h.resize_input:
            self.unet = Unet(channels_in=3, channels_out=1)
        else:
            self.unet = UNetMini(channels_in=3, channels_out=1)

    def configure_optimizers(self):
        h = self.hparams
        groups = []
        if self.unet is not None:
            groups.append({"params": self.unet.parameters(), "lr": h.lr})
        groups += [{"params": self.project.parameters(), "lr": 10 * h.lr}, {"params": self.ifnet.parameters()}]
        return [torch.optim.Adam(groups, lr=h.lr)], []

    def forward(self, batch):
        h = self.hparams
        if self.unet is not None:
            raw = self.unet(batch["rgb"])
            if h.resize_input:
                logits = F.interpolate(raw, size=320, mode="bilinear")[:, :, 40:280, :].squeeze(1)
            else:
                logits = raw
            depth = torch.sigmoid(logits) * (h.max_z - h.min_z) + h.min_z
        else:
            depth = batch["depthmap_target"]
        # unproject + normalise fused in one kernel
        point_cloud = self.project.depthmap_to_gridspace(depth.contiguous(), h.scale_factor, normalize=True)
        voxel_occupancy = self.project(point_cloud)
        # trainer_scene_net.py:91-99.  The reference's first condition reads `n < (240*320) & n > 0`: `&` binds tighter
        # than the comparisons, so it is the chain  n < (76800 & n) > 0 , which no n satisfies (76800 & n <= n) -- the
        # random-subset branch is dead code there and the whole point cloud is used whenever n != 0.  Mirrored as is.
        n = h.subsample_points
        masked = (240 * 320) & n if n > 0 else 0
        if n < masked and masked > 0:
            indices = torch.randperm(point_cloud.shape[1], device=point_cloud.device)[:n]
            point_cloud = point_cloud[:, indices, :].contiguous()
            points = torch.cat((point_cloud, batch["points"]), dim=1)
        elif n == 0:
            points = batch["points"]
        else:
            points = torch.cat((point_cloud, batch["points"]), dim=1)
        logits_depth = self.ifnet(voxel_occupancy, points)
        return logits_depth, depth, point_cloud

    def _occupancies(self, batch, point_cloud):
        """trainer_scene_net.py:108-114: ground truth of the extra query points = on-the-fly labelling against the mesh."""
        if self.hparams.subsample_points == 0:
            return batch["occupancies"]
        # (the reference calls it with the default dims (139, 104, 112) whatever scale_factor is, :112)
        _, occ_pc = determine_occupancy(batch["mesh"], point_cloud.detach(),
                                        reference_quirk=getattr(self.hparams, "reference_occupancy_quirk", True),
                                        points_normalized=True)
        return torch.cat((occ_pc.to(batch["occupancies"].dtype), batch["occupancies"]), dim=1)

    def losses_and_logging(self, batch, depthmap, logits, occupancies, mode="train"):
        ce_loss = _BCELogitsMeanFn.apply(logits, occupancies)
        mse_loss = F.mse_loss(depthmap, batch["depthmap_target"], reduction="mean")
        mesh_ce_loss = ce_loss
        if self.hparams.subsample_points > 0:          # :151-154 (logged only)
            k = self.hparams.subsample_points
            mesh_ce_loss = _BCELogitsMeanFn.apply(logits[:, k:].contiguous(), occupancies[:, k:].contiguous())
        self.last_log = {f"{mode}_ce_loss": ce_loss.detach(), f"{mode}_mse_depth_loss": mse_loss.detach(),
                         f"{mode}_mesh_ce_loss": mesh_ce_loss.detach(),
                         "sigma_x": self.project.sigma[2].detach(), "sigma_y": self.project.sigma[1].detach(),
                         "sigma_z": self.project.sigma[0].detach()}
        if self.hparams.no_depth_sup:
            return ce_loss
        return ce_loss + mse_loss

    def training_step(self, batch, batch_idx):
        logits, depthmap, point_cloud = self.forward(batch)
        occupancies = self._occupancies(batch, point_cloud)
        loss = self.losses_and_logging(batch, depthmap, logits, occupancies, "train")
        return {"loss": loss}
