"""Prompt templating of the Zephyr chat format (API of ref:vis_zephyr/conversation.py:9-125).

Pure host-side string work, unchanged in meaning: `<|system|>\\n{system}</s><|user|>\\n{msg}</s><|assistant|>\\n`.
"""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional, Sequence


class SeparatorStyle(Enum):
    ZEPHYR = auto()
    PLAIN = auto()


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Sequence[str]
    messages: Sequence[Sequence[str]]
    offset: int
    separator_style: SeparatorStyle = SeparatorStyle.ZEPHYR
    separator_01: str = "</s>"
    separator_02: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def get_prompt(self) -> str:
        if self.separator_style != SeparatorStyle.ZEPHYR:
            raise ValueError(f"Unknown separator style: {self.separator_style}")
        end = self.separator_01
        parts = [f"<|system|>\n{self.system}{end}"]
        for role, message in self.messages:
            if isinstance(message, tuple):          # (text, image, mode) triples keep only the text
                message = message[0]
            parts.append(f"<|{role}|>\n{message}{end}" if message else f"<|{role}|>\n")
        return "".join(parts)

    def append_message(self, role, message):
        if not isinstance(self.messages, list):
            self.messages = [list(m) for m in self.messages]
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages],
                            offset=self.offset, separator_style=self.separator_style, separator_01=self.separator_01,
                            separator_02=self.separator_02, version=self.version)


_VCR_SYSTEM_V1 = ("You are an AI assistant specialized in Visual Commonsense Reasoning and able to understand the visual "
                  "content that the user provides.\nGiven an image and a question, your task is to provide an accurate "
                  "answer, followed by a concise, logical explanation of your reasoning based on visual cues and common "
                  "sense. Your response must clearly separate the answer and the explanation.")
_VCR_SYSTEM_MC = ("You are an AI assistant specialized in Visual Commonsense Reasoning. Your task is to analyze the provided "
                  "visual content along with a question. Subsequently, select the most appropriate answer from the given "
                  "choices. Your answer must be in the format 'Answer is: {A, B, C or D}'.")

conv_zephyr_v1 = Conversation(system=_VCR_SYSTEM_V1, roles=("user", "assistant"), messages=(), offset=0, version="zephyr_v1")
conv_zephyr_vcr = Conversation(system=_VCR_SYSTEM_MC, roles=("user", "assistant"), messages=(), offset=0, version="zephyr_vcr")
conv_zephyr_plain = Conversation(system="", roles=("", ""), messages=(), offset=0, separator_style=SeparatorStyle.PLAIN,
                                 version="plain")

default_conversation = conv_zephyr_v1
templates = {"default": conv_zephyr_v1, "zephyr_v1": conv_zephyr_v1, "zephyr_vcr": conv_zephyr_vcr, "plain": conv_zephyr_plain}
