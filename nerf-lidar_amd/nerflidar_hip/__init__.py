"""nerflidar_hip: host-side mirror of the NeRF-LiDAR zipnerf interface over libnerflidar_hip.so (HIP, gfx950).

Modules: gridencoder, models, lidar, camera, sharding, checkpoints, render_lidar, raydrop, objects, training, config, weights.
"""
__version__ = "0.1.0"
