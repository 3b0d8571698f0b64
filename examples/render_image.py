"""Render the built-in scene on the GPU and write it as TGA (the reference's Cs/Program.cs:33-59 format) and PNG.

    python examples/render_image.py [width height frames]
"""
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from toypathtracer_amd import api  # noqa: E402


def write_png(path, rgba):
    h, w = rgba.shape[:2]
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 720
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    out_dir = sys.argv[4] if len(sys.argv) > 4 else "."
    api.InitializeTest()
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    r0 = api.ray_counter_read()
    for f in range(frames):
        api.UpdateTest(0.0, f, w, h, api.kFlagProgressive)
        api.draw_device(0.0, f, w, h, tile.data_ptr(), api.kFlagProgressive)
    api.display_rgba8(tile.data_ptr(), w, h, rgba.data_ptr())
    rays = api.ray_counter_read() - r0
    img = rgba.cpu().numpy()
    api.write_tga(os.path.join(out_dir, "output.tga"), img)
    write_png(os.path.join(out_dir, "output.png"), img)
    print("%dx%d, %d frames x 4 spp, %d rays -> output.tga / output.png" % (w, h, frames, rays))
    api.ShutdownTest()


if __name__ == "__main__":
    main()
