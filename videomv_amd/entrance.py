"""``inference_text2video_entrance`` — the INFER_ENGINE plugin of ``inference.py --cfg configs/t2v_infer.yaml``
(tools/inferences/inference_text2video_entrance.py:37-328) over the HIP hot path.

Kept: config layering (python defaults <- YAML/CLI dict-merge <- ``vldm_cfg`` overlay), one process per GPU with
``seed + rank`` (replicas), registry-built DIFFUSION / EMBEDDER / AUTO_ENCODER / MODEL, checkpoint formats
(``state_dict``/``step`` wrapper, ``strict=False``), orbit ``camera_data``, latent noise shape
``[1, 4, max_frames, res_y/scale, res_x/scale]``, CFG kwargs pair, 50-step DDIM (``cfg.ddim_timesteps``, new key — the
reference hard-codes 50 at :264), VAE decode in ``decoder_bs`` chunks, output naming.
Different by design: launches go to ``libvmv_hip_{f16,bf16}.so`` (``hip_dtype`` config key / ``VMV_DTYPE``); per sample the entrance
writes the reference's frame PNGs (``<name>/{fid:05d}.png``) and ``<name>.mp4`` when an H.264 encoder exists (none ships here),
plus a ``.pt`` tensor and a PNG contact sheet; the second, LGM-refined loop (:271-278)
runs when ``UNet.use_lgm_refine`` is set (also under ``frame_parallel``; not under ``cfg_parallel``).
"""
import logging
import os
import os.path as osp
import re
import sys

import torch
import torch.distributed as dist

from .config import default_cfg, merge_into, assign_signle_cfg, AttrDict
from .registry import INFER_ENGINE, MODEL, EMBEDDER, AUTO_ENCODER, DIFFUSION
from .camera import entrance_camera_data
from .pipeline import sample_views
from .dist import rank_seed, shard_prompts
from . import embedder as _embedder  # noqa: F401  (registers the embedders)


def _plain(d):
    return {k: (_plain(v) if isinstance(v, dict) else v) for k, v in d.items()}


def _load_weights(module, path, allow_random, what, prefix_filter=None):
    if path and osp.exists(path):
        sd = torch.load(path, map_location="cpu")
        if "state_dict" in sd:
            sd = sd["state_dict"]
        if prefix_filter:
            sd = {k.split(prefix_filter)[-1]: v for k, v in sd.items() if prefix_filter in k}
        status = module.load_state_dict(sd, strict=False)
        logging.info(f"Load {what} from {path} with status {status}")
        return True
    if not allow_random:
        raise FileNotFoundError(f"{what} checkpoint {path!r} not found (set allow_random_init True to run on "
                                f"random weights)")
    logging.warning(f"{what}: checkpoint {path!r} missing -> RANDOM weights (allow_random_init)")
    g = torch.Generator().manual_seed(1234)
    for name, p in module.named_parameters():
        if p.dim() > 1:
            p.copy_(torch.randn(p.shape, generator=g) * (p[0].numel() ** -0.5))
        elif name.endswith("bias"):
            p.zero_()
        else:
            p.fill_(1.0)
    return False


@INFER_ENGINE.register_function()
def inference_text2video_entrance(cfg_update, **kwargs):
    cfg = default_cfg()
    merge_into(cfg, _plain(dict(cfg_update)))
    cfg.pmi_rank = int(os.getenv('RANK', 0))
    cfg.pmi_world_size = int(os.getenv('WORLD_SIZE', 1))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    cfg.world_size = 1 if cfg.debug else cfg.pmi_world_size       # launched one rank per GPU by torch.distributed.run
    return worker(int(os.getenv('LOCAL_RANK', 0)), cfg, cfg_update)   # (returns the worker's merged cfg)


def _warn_bf16(name):
    """bf16 is the range fallback, not a parity configuration (configs/*.yaml): say so once, where the user chose it."""
    if str(name).lower() in ("bf16", "bfloat16"):
        import warnings
        warnings.warn("hip_dtype: bf16 — the wide-range build; measured 1.3e-2 rel-L2 per UNet forward against the fp32 reference, outside "
                      "the 1e-2 parity tolerance that the default fp16 build meets (use it only for checkpoints that overflow fp16)")


@torch.no_grad()
def worker(gpu, cfg, cfg_update):
    if 'vldm_cfg' in cfg_update and cfg_update['vldm_cfg']:
        cfg = AttrDict(assign_signle_cfg(cfg, cfg_update, 'vldm_cfg'))
        merge_into(cfg, _plain(dict(cfg_update)))
    cfg.gpu, cfg.seed = gpu, int(cfg.seed)
    if cfg.get('hip_dtype'):                        # (not a reference key) 16-bit storage type of the kernels: fp16 | bf16
        from . import _lib
        _lib.set_elem(cfg.hip_dtype)
        _warn_bf16(cfg.hip_dtype)
    cfg.rank = cfg.pmi_rank
    # frame_parallel (not a reference key; BASELINE configs[2]): the ranks share ONE sample — same seed, F / N views each
    fpar = bool(cfg.get('frame_parallel', False)) and cfg.world_size > 1
    torch.manual_seed(cfg.seed if fpar else rank_seed(cfg.seed, cfg.rank))
    on_gpu = str(cfg.device).startswith("cuda")
    device = torch.device("cuda", gpu) if on_gpu else torch.device(cfg.device)
    if on_gpu:
        torch.cuda.set_device(gpu)
    if cfg.world_size > 1:
        dist.init_process_group(backend='nccl' if on_gpu else 'gloo', world_size=cfg.world_size, rank=cfg.rank)

    exp_name = osp.basename(cfg.test_list_path).split('.')[0]
    cfg.log_dir = osp.join(cfg.log_dir, exp_name)
    os.makedirs(cfg.log_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s] %(levelname)s: %(message)s', force=True,
                        handlers=[logging.FileHandler(osp.join(cfg.log_dir, 'log_%02d.txt' % cfg.rank)),
                                  logging.StreamHandler(stream=sys.stdout)])
    logging.info(f"Going into inference_text2video_entrance inference on {gpu} gpu (HIP hot path)")

    diffusion = DIFFUSION.build(dict(cfg.Diffusion))
    clip_encoder = EMBEDDER.build(dict(cfg.embedder))
    _, _, zero_y_negative = clip_encoder(text=[""])
    autoencoder = AUTO_ENCODER.build({k: v for k, v in cfg.auto_encoder.items() if k != 'pretrained'})
    _load_weights(autoencoder, cfg.auto_encoder.get('pretrained'), cfg.allow_random_init, "autoencoder",
                  prefix_filter='first_stage_model.')
    autoencoder.eval()
    unet_cfg = dict(cfg.UNet)
    # (LGM-refined steps run frame-parallel too since round 4 — every rank renders / re-encodes its own views; not with cfg_parallel)
    use_lgm = bool(unet_cfg.get('use_lgm_refine')) and not (fpar and bool(cfg.get('cfg_parallel', False)))
    unet_cfg['use_lgm_refine'] = use_lgm
    if use_lgm and cfg.get('lgm_opt'):              # (not a reference key: shrinks the LGM for CPU plumbing tests)
        unet_cfg['lgm_opt'] = _plain(dict(cfg.lgm_opt))
    model = MODEL.build(unet_cfg)
    _load_weights(model, cfg.get('test_model'), cfg.allow_random_init, "UNet")
    model.eval()
    if fpar:
        from .comm import FrameComm, CfgFrameComm
        # cfg_parallel (not a reference key): 2 x N/2 — one CFG branch per half of the ranks, frames sharded inside each half
        model.set_frame_parallel(CfgFrameComm() if bool(cfg.get('cfg_parallel', False)) else FrameComm())
        logging.info(f"frame-parallel sampling: {cfg.world_size} ranks x {int(cfg.num_views or cfg.max_frames) // cfg.world_size} views")

    with open(cfg.test_list_path, 'r') as f:
        test_list = [ln.strip() for ln in f.readlines()]
    # `shard_prompts: True` (not a reference key): rank r takes prompts r, r + W, ... instead of every rank running the whole list
    # with seed + rank (the reference's behaviour, :152-156, and the default)
    if not fpar:
        test_list = shard_prompts(test_list, cfg.rank, max(1, int(cfg.get('world_size', 1))), replicate=not cfg.get('shard_prompts', False))
    F = int(cfg.num_views or cfg.max_frames)
    lat_h, lat_w = int(cfg.resolution[1] / cfg.scale), int(cfg.resolution[0] / cfg.scale)
    outputs = []
    # `prompt_batch: b` (not a reference key; default 1 = the reference's one prompt at a time, :152-214): b prompts are denoised in ONE
    # plan per step (noise [b, 4, F, h, w], y [b, 77, 1024] — the shapes the reference's sampler API admits, diffusion_ddim.py:247-260).
    # One sample's small levels do not fill 256 CUs: 2 prompts per plan = +22 % samples/s at 256 px, +6 % at 320 x 512.
    # Noises are drawn per prompt in list order, so every sample starts from the noise the unbatched run gives it (without the LGM-refined
    # loop; with it the refined steps' posterior draws sit between two prompts' noises in the unbatched run, so the noises — not their
    # distribution — differ).  The LGM-refined second loop is batched too: its 47 plain steps in one plan, the 3 refined ones sample by
    # sample (diffusion_ddim.ddim_sample_loop).  Over a frame-parallel group (plain `frame_parallel`, no LGM loop) every rank denoises its
    # frames of all b prompts in one plan.
    # (the fused LGM steps are GPU-only; over a frame-parallel group: plain FrameComm without the LGM loop — DESIGN 8)
    one_only = (use_lgm and (not on_gpu or fpar)) or (fpar and bool(cfg.get('cfg_parallel', False)))
    pbatch = 1 if one_only else max(1, int(cfg.get('prompt_batch', 1) or 1))
    elevation, camera_dist = 15, 2.0

    def run_group(group):
        camera_data = entrance_camera_data(F, elevation=elevation, camera_distance=camera_dist)
        ys = [clip_encoder(text=[cap])[2] for _, cap in group]
        noises = [torch.randn([1, 4, F, lat_h, lat_w]) for _ in group]
        y_words, noise = torch.cat(ys, dim=0), torch.cat(noises, dim=0).to(device)
        y_neg = zero_y_negative.to(device).expand(len(group), -1, -1).contiguous()      # (the reference's forward wants y of the noise's batch)
        x0_all, video_all = sample_views(model, diffusion, autoencoder, noise, y_words.to(device), y_neg,
                                         camera_data, guide_scale=cfg.guide_scale, ddim_timesteps=int(cfg.ddim_timesteps),
                                         decoder_bs=int(cfg.decoder_bs), scale_factor=cfg.scale_factor)
        x0_gs_all = video_gs_all = None
        if use_lgm:       # second, LGM-refined loop from the SAME noise (inference_text2video_entrance.py:267-279,302-311)
            from .lgm import prepare_gs_data
            from .pipeline import decode_views
            gs_data = prepare_gs_data(camera_data, model.lgm_opt)
            kw = [dict(y=y_words.to(device), camera_data=camera_data, gs_data=gs_data),
                  dict(y=y_neg, camera_data=camera_data, gs_data=gs_data)]
            x0_gs_all = diffusion.ddim_sample_loop(noise=noise, model=model, autoencoder=autoencoder, model_kwargs=kw,
                                                   guide_scale=cfg.guide_scale, ddim_timesteps=int(cfg.ddim_timesteps), eta=0.0)
            video_gs_all = decode_views(autoencoder, x0_gs_all, int(cfg.decoder_bs), cfg.scale_factor)
        if fpar and cfg.rank != 0:          # every rank holds the gathered views; rank 0 writes them
            return
        for s, (idx, caption) in enumerate(group):
            x0, video = x0_all[s:s + 1], video_all[s:s + 1]
            x0_gs, video_gs = (None, None) if video_gs_all is None else (x0_gs_all[s:s + 1], video_gs_all[s:s + 1])
            cap_name = re.sub(r'[^\w\s]', '', caption).replace(' ', '_')
            stem = f'rank_{cfg.world_size:02d}_{cfg.rank:02d}_{idx:04d}_{cap_name}_{int(elevation):02d}_{camera_dist:.02f}'
            path = osp.join(cfg.log_dir, stem + '.pt')
            torch.save({'latent': x0.cpu(), 'video': video.cpu(), 'caption': caption}, path)
            _save_contact_sheet(video.cpu(), osp.join(cfg.log_dir, stem + '.png'), cfg.mean, cfg.std)
            _save_frames_safe(osp.join(cfg.log_dir, stem + '.mp4'), video, cfg)          # the reference's <name>.mp4 + frame PNGs
            if video_gs is not None:                     # the reference's second file: <name>_gs
                torch.save({'latent': x0_gs.cpu(), 'video': video_gs.cpu(), 'caption': caption}, osp.join(cfg.log_dir, stem + '_gs.pt'))
                _save_contact_sheet(video_gs.cpu(), osp.join(cfg.log_dir, stem + '_gs.png'), cfg.mean, cfg.std)
                _save_frames_safe(osp.join(cfg.log_dir, stem + '_gs.mp4'), video_gs, cfg)
            logging.info('Save views to %s' % path)
            outputs.append(path)

    pending = []
    for idx, caption in enumerate(test_list):
        if caption.startswith('#') or caption == "":
            logging.info(f'Skip {caption!r}')
            continue
        if '3d asset' not in caption:
            caption = caption + ", 3d asset"
        logging.info(f"[{idx}]/[{len(test_list)}] Begin to sample {caption} ...")
        pending.append((idx, caption))
        if len(pending) == pbatch:
            run_group(pending)
            pending = []
    if pending:
        run_group(pending)
    logging.info('Congratulations! The inference is completed!')
    if on_gpu:
        torch.cuda.synchronize()
    if cfg.world_size > 1:
        fc = getattr(getattr(model, "module", model), "frame_comm", None)
        if fc is not None and hasattr(fc, "close"):
            fc.close()              # the native RCCL communicators (and twins) go first: collective, every rank gets here
        dist.barrier()
        dist.destroy_process_group()
    cfg.outputs = outputs
    return cfg


def center_crop_wide(img, size):
    """utils/transforms.py:163-183 — BOX-resize so that the image covers ``size`` (w, h), then centre-crop."""
    from PIL import Image
    scale = min(img.size[0] / size[0], img.size[1] / size[1])
    img = img.resize((round(img.width // scale), round(img.height // scale)), resample=Image.BOX)
    x1, y1 = (img.width - size[0]) // 2, (img.height - size[1]) // 2
    return img.crop((x1, y1, x1 + size[0], y1 + size[1]))


def _to_normalised_tensor(img, mean, std):
    import numpy as np
    a = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (a - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)


@INFER_ENGINE.register_function()
def inference_i2vgen_entrance(cfg_update, **kwargs):
    """Image -> 24 views (tools/inferences/inference_i2vgen_entrance.py): RGBA image on a white background -> VAE-encoded
    ``local_image`` + CLIP image feature -> UNetSD_I2VGen with v-prediction / cosine-ZTSNR DDIM, guide 6, orbit cameras
    at elevation 5 / distance 1.7 (:136-147), fps 8; decode in ``decoder_bs`` chunks."""
    cfg = default_cfg()
    cfg.negative_prompt = 'Distorted, discontinuous, Ugly, blurry, low resolution, motionless, static, disfigured, ' \
                          'disconnected limbs, Ugly faces, incomplete arms'
    merge_into(cfg, _plain(dict(cfg_update)))
    cfg.pmi_rank, cfg.pmi_world_size = int(os.getenv('RANK', 0)), int(os.getenv('WORLD_SIZE', 1))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    cfg.world_size = 1 if cfg.debug else cfg.pmi_world_size
    return worker_i2v(int(os.getenv('LOCAL_RANK', 0)), cfg, cfg_update)


@torch.no_grad()
def worker_i2v(gpu, cfg, cfg_update):
    from PIL import Image
    if 'vldm_cfg' in cfg_update and cfg_update['vldm_cfg']:
        cfg = AttrDict(assign_signle_cfg(cfg, cfg_update, 'vldm_cfg'))
        merge_into(cfg, _plain(dict(cfg_update)))
    cfg.gpu, cfg.seed, cfg.rank = gpu, int(cfg.seed), cfg.pmi_rank
    if cfg.get('hip_dtype'):                        # (not a reference key) 16-bit storage type of the kernels: fp16 | bf16
        from . import _lib
        _lib.set_elem(cfg.hip_dtype)
        _warn_bf16(cfg.hip_dtype)
    torch.manual_seed(rank_seed(cfg.seed, cfg.rank))
    on_gpu = str(cfg.device).startswith("cuda")
    device = torch.device("cuda", gpu) if on_gpu else torch.device(cfg.device)
    if on_gpu:
        torch.cuda.set_device(gpu)
    if cfg.world_size > 1:
        dist.init_process_group(backend='nccl' if on_gpu else 'gloo', world_size=cfg.world_size, rank=cfg.rank)
    exp_name = osp.basename(cfg.test_list_path).split('.')[0]
    cfg.log_dir = osp.join(cfg.log_dir, exp_name)
    os.makedirs(cfg.log_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s] %(levelname)s: %(message)s', force=True,
                        handlers=[logging.FileHandler(osp.join(cfg.log_dir, 'log_%02d.txt' % cfg.rank)),
                                  logging.StreamHandler(stream=sys.stdout)])
    logging.info(f"Going into inference_i2vgen_entrance inference on {gpu} gpu (HIP hot path)")
    diffusion = DIFFUSION.build(dict(cfg.Diffusion))
    clip_encoder = EMBEDDER.build(dict(cfg.embedder))
    _, _, zero_y_negative = clip_encoder(text=cfg.negative_prompt)
    black_image_feature = torch.zeros([1, 1, cfg.UNet['y_dim']])
    autoencoder = AUTO_ENCODER.build({k: v for k, v in cfg.auto_encoder.items() if k != 'pretrained'})
    _load_weights(autoencoder, cfg.auto_encoder.get('pretrained'), cfg.allow_random_init, "autoencoder",
                  prefix_filter='first_stage_model.')
    autoencoder.eval()
    unet_cfg = dict(cfg.UNet)
    use_lgm = bool(unet_cfg.get('use_lgm_refine'))
    if use_lgm and cfg.get('lgm_opt'):              # (not a reference key: shrinks the LGM for CPU plumbing tests)
        unet_cfg['lgm_opt'] = _plain(dict(cfg.lgm_opt))
    model = MODEL.build(unet_cfg)
    _load_weights(model, cfg.get('test_model'), cfg.allow_random_init, "UNet")
    model.eval()
    F = int(cfg.num_views or cfg.max_frames)
    elevation, camera_dist = 5, 1.7
    camera_data = entrance_camera_data(F, elevation=elevation, camera_distance=camera_dist)
    with open(cfg.test_list_path, 'r') as f:
        test_list = [ln.strip() for ln in f.readlines() for _ in range(int(cfg.get('round', 1)))]
    lat_h, lat_w = int(cfg.resolution[1] / cfg.scale), int(cfg.resolution[0] / cfg.scale)
    outputs = []
    # `prompt_batch: b` (not a reference key; default 1 = the reference's one image at a time, inference_i2vgen_entrance.py:215-300): b input
    # images are denoised in ONE plan per step (unet_i2vgen._forward_cfg_rows_batched): +23 % samples/s at 256 px, +7 % at 320 x 512 with
    # b = 2.  Noises are drawn per image in list order; the LGM-refined second loop is batched too (47 plain steps in one plan, the 3
    # refined ones sample by sample: diffusion_ddim.ddim_sample_loop).
    pbatch = 1 if (use_lgm and not on_gpu) else max(1, int(cfg.get('prompt_batch', 1) or 1))

    def run_group(group):
        ys, vis, locs, noises = [], [], [], []
        for idx, line, image in group:
            vit_img = center_crop_wide(image, (cfg.resolution[0], cfg.resolution[0])).resize(tuple(cfg.get('vit_resolution', [224, 224])))
            y_visual, _, y_words = clip_encoder(image=_to_normalised_tensor(vit_img, cfg.mean, cfg.std).unsqueeze(0), text=[""])
            img_t = _to_normalised_tensor(center_crop_wide(image, tuple(cfg.resolution)), cfg.mean, cfg.std).unsqueeze(0).to(device)
            local_image = autoencoder.encode_firsr_stage(img_t, cfg.scale_factor)
            ys.append(y_words.to(device)); vis.append(y_visual.unsqueeze(1).to(device))
            locs.append(local_image.unsqueeze(2).repeat_interleave(repeats=F, dim=2))
            noises.append(torch.randn([1, 4, F, lat_h, lat_w]))
        n = len(group)
        y_words, y_visual, local_image = torch.cat(ys, dim=0), torch.cat(vis, dim=0), torch.cat(locs, dim=0)
        noise = torch.cat(noises, dim=0).to(device)
        fps_tensor = torch.tensor([cfg.target_fps], dtype=torch.long, device=device)
        infer_img = black_image_feature if cfg.use_zero_infer else None
        rep = lambda v: v.to(device).expand(n, *v.shape[1:]).contiguous()        # (the reference's forward wants kwargs of the noise's batch)
        kw = [{'y': y_words, 'image': y_visual, 'local_image': local_image, 'fps': fps_tensor, 'camera_data': camera_data},
              {'y': rep(zero_y_negative), 'image': None if infer_img is None else rep(infer_img),
               'local_image': local_image, 'fps': fps_tensor, 'camera_data': camera_data}]
        x0_all = diffusion.ddim_sample_loop(noise=noise, model=model, model_kwargs=kw, guide_scale=cfg.guide_scale,
                                            ddim_timesteps=int(cfg.ddim_timesteps), eta=0.0)
        from .pipeline import decode_views
        video_all = decode_views(autoencoder, x0_all, int(cfg.decoder_bs), cfg.scale_factor)
        x0_gs_all = video_gs_all = None
        if use_lgm:       # second, LGM-refined loop from the same noise (inference_i2vgen_entrance.py:281-292)
            from .lgm import prepare_gs_data
            gs_data = prepare_gs_data(camera_data, model.lgm_opt)
            kw_gs = [dict(k, gs_data=gs_data) for k in kw]
            x0_gs_all = diffusion.ddim_sample_loop(noise=noise, model=model, autoencoder=autoencoder, model_kwargs=kw_gs,
                                                   guide_scale=cfg.guide_scale, ddim_timesteps=int(cfg.ddim_timesteps), eta=0.0)
            video_gs_all = decode_views(autoencoder, x0_gs_all, int(cfg.decoder_bs), cfg.scale_factor)
        for s_, (idx, line, _) in enumerate(group):
            x0, video = x0_all[s_:s_ + 1], video_all[s_:s_ + 1]
            x0_gs, video_gs = (None, None) if video_gs_all is None else (x0_gs_all[s_:s_ + 1], video_gs_all[s_:s_ + 1])
            stem = f'rank_{cfg.world_size:02d}_{cfg.rank:02d}_{idx:04d}_{osp.basename(line).split(".")[0]}_{int(elevation):02d}_{camera_dist:.02f}'
            path = osp.join(cfg.log_dir, stem + '.pt')
            torch.save({'latent': x0.cpu(), 'video': video.cpu(), 'image': line}, path)
            _save_contact_sheet(video.cpu(), osp.join(cfg.log_dir, stem + '.png'), cfg.mean, cfg.std)
            _save_frames_safe(osp.join(cfg.log_dir, stem + '.mp4'), video, cfg)
            if video_gs is not None:
                torch.save({'latent': x0_gs.cpu(), 'video': video_gs.cpu(), 'image': line}, osp.join(cfg.log_dir, stem + '_gs.pt'))
                _save_contact_sheet(video_gs.cpu(), osp.join(cfg.log_dir, stem + '_gs.png'), cfg.mean, cfg.std)
                _save_frames_safe(osp.join(cfg.log_dir, stem + '_gs.mp4'), video_gs, cfg)
            logging.info('Save views to %s' % path)
            outputs.append(path)

    pending = []
    for idx, line in enumerate(test_list):
        if line.startswith('#') or line == "":
            logging.info(f'Skip {line!r}')
            continue
        try:
            rgba = Image.open(line).convert('RGBA')
        except Exception as e:
            logging.info(f'cannot open {line}: {e}')
            continue
        logging.info(f"[{idx}]/[{len(test_list)}] Begin to sample {line} ...")
        image = Image.new('RGB', size=rgba.size, color=(255, 255, 255))
        image.paste(rgba, (0, 0), mask=rgba)
        pending.append((idx, line, image))
        if len(pending) == pbatch:
            run_group(pending)
            pending = []
    if pending:
        run_group(pending)
    logging.info('Congratulations! The inference is completed!')
    if on_gpu:
        torch.cuda.synchronize()
    if cfg.world_size > 1:
        fc = getattr(getattr(model, "module", model), "frame_comm", None)
        if fc is not None and hasattr(fc, "close"):
            fc.close()              # the native RCCL communicators (and twins) go first: collective, every rank gets here
        dist.barrier()
        dist.destroy_process_group()
    cfg.outputs = outputs
    return cfg


def _save_frames_safe(local_path, video, cfg):
    try:                                            # (inference_text2video_entrance.py:296-300: errors are logged, not raised)
        save_video_frames(local_path, video, cfg.mean, cfg.std)
        logging.info('Save video to dir %s:' % local_path)
    except Exception as e:
        logging.info(f'Step: save text or video error with {e}')


def save_video_frames(local_path, gen_video, mean, std, save_fps=8):
    """The reference writer's outputs (utils/video_op.py:166-211, ``save_i2vgen_video_safe``) as far as this image allows:
    the de-normalised frames ``clamp(v * std + mean, 0, 1) * 255`` truncated to uint8, one PNG per view named
    ``<local_path minus .mp4>/{fid:05d}.png`` (written by cv2 in the reference, PIL here: same pixels), and
    ``<local_path>`` itself — an H.264 mp4 at ``save_fps`` — when an encoder exists (``imageio`` with ffmpeg, or an ``ffmpeg``
    binary on PATH; neither ships in this image, in which case only the PNGs are written and the fact is logged).
    A single-frame video becomes ``<local_path>.png`` as in the reference.  Returns the list of files written."""
    import shutil
    import subprocess
    import numpy as np
    from PIL import Image
    v = gen_video.detach().float().cpu()
    v = v * torch.tensor(std).view(1, -1, 1, 1, 1) + torch.tensor(mean).view(1, -1, 1, 1, 1)
    v = (v.clamp(0, 1) * 255.0)[0].permute(1, 2, 3, 0)                   # f h w c  (the reference saves sample 0)
    frames = [f.numpy().astype('uint8') for f in v]                      # astype truncates, as the reference does
    written = []
    if len(frames) == 1:
        Image.fromarray(frames[0]).save(local_path + '.png')
        return [local_path + '.png']
    frame_dir = local_path.replace('.mp4', '')
    os.makedirs(frame_dir, exist_ok=True)
    for fid, frame in enumerate(frames):
        fp = os.path.join(frame_dir, '{:05d}.png'.format(fid))
        Image.fromarray(frame).save(fp)
        written.append(fp)
    try:
        import imageio                                                    # the reference's encoder
        writer = imageio.get_writer(local_path, fps=save_fps, codec='libx264', quality=8)
        for frame in frames:
            writer.append_data(frame)
        writer.close()
        written.append(local_path)
    except Exception:
        exe = shutil.which('ffmpeg')
        if exe is not None:
            cmd = [exe, '-y', '-loglevel', 'quiet', '-framerate', str(save_fps), '-start_number', '0', '-i',
                   os.path.join(frame_dir, '%05d.png'), '-vcodec', 'libx264', '-crf', '17', '-pix_fmt', 'yuv420p', local_path]
            if subprocess.run(cmd).returncode == 0:
                written.append(local_path)
        else:
            logging.info(f'no H.264 encoder in this environment (imageio / ffmpeg): wrote {len(frames)} PNG frames to {frame_dir}')
    return written


def _save_contact_sheet(video, path, mean, std):
    """video [1, 3, F, H, W] in normalised range -> one PNG with the F views side by side (best effort)."""
    try:
        from PIL import Image
        v = video[0].permute(1, 2, 3, 0).float()                     # F H W C
        v = (v * torch.tensor(std) + torch.tensor(mean)).clamp(0, 1)
        sheet = torch.cat(list(v), dim=1)                            # H, F*W, C
        Image.fromarray((sheet * 255).round().byte().numpy()).save(path)
    except Exception as e:  # pragma: no cover
        logging.info(f'Step: save png error with {e}')
