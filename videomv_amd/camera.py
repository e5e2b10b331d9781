"""Orbit cameras for the 24 views (utils/camera_utils.py:4-61) and the entrance's row manipulation
(inference_text2video_entrance.py:184-191) -> ``camera_data`` [1, F, 16].  Host-side numpy, once per prompt."""
import numpy as np
import torch


def camera_to_world(elevation_deg, azimuth_deg, distance=1.0):
    el, az = np.radians(elevation_deg), np.radians(azimuth_deg)
    pos = distance * np.array([np.cos(el) * np.sin(az), np.sin(el), np.cos(el) * np.cos(az)])
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    up /= np.linalg.norm(up)
    m = np.eye(4)
    m[:3, :3] = np.stack([right, up, -fwd], axis=1)
    m[:3, 3] = pos
    return m


def get_camera(num_frames, elevation=15, azimuth_start=0, azimuth_span=360, blender_coord=True, camera_distance=1.0):
    flip_yz = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    cams = []
    for az in np.arange(azimuth_start, azimuth_span + azimuth_start, azimuth_span / num_frames):
        m = camera_to_world(elevation, az, camera_distance)
        if blender_coord:
            m = flip_yz @ m
        cams.append(m.flatten())
    return torch.tensor(np.stack(cams, 0)).float()


def entrance_camera_data(num_frames=24, elevation=15, camera_distance=2.0):
    cam = get_camera(num_frames, elevation=elevation, azimuth_start=0, azimuth_span=360,
                     camera_distance=camera_distance).unsqueeze(0).reshape(1, num_frames, 4, 4)
    cam[:, :, 1, :] *= -1
    cam[:, :, [0, 1], :] = cam[:, :, [1, 0], :]
    return cam.reshape(1, num_frames, 16)
