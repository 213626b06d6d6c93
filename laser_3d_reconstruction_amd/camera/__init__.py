from .single_usb_stereo_camera import SingleUSBStereoCameraManager

__all__ = ['SingleUSBStereoCameraManager']
