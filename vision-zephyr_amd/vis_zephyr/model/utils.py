"""ref:vis_zephyr/model/utils.py:23-28."""


def preprocess_image(image_path, image_processor):
    from PIL import Image
    return image_processor(images=Image.open(image_path).convert("RGB"), return_tensors="pt")["pixel_values"]
