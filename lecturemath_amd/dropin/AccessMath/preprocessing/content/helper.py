"""Helper.decompress_binary_images (content/helper.py:27-34): PNG list -> list of uint8 frames."""
from lecturemath_amd import png


class Helper:
    @staticmethod
    def decompress_binary_images(compressed_images):
        return [png.decode_gray8(raw) for raw in compressed_images]
