"""The wire format of the reference's gRPC service (src/serving/grpc/proto/llm.proto:1-86), built as runtime descriptors:
there is no protoc in this image, and the message layout (field numbers and types) IS the compatibility contract.

    service ppl.llm.proto.LLMService { rpc Generation (BatchedRequest) returns (stream BatchedResponse) }
"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto
PKG = "ppl.llm.proto"


def _msg(fd, name, fields):
    m = fd.message_type.add()
    m.name = name
    for fname, num, ftype, label, tname in fields:
        f = m.field.add()
        f.name, f.number, f.type, f.label = fname, num, ftype, label
        if tname:
            f.type_name = f".{PKG}.{tname}"
    return m


def _build():
    fd = descriptor_pb2.FileDescriptorProto(name="ppl_llm.proto", package=PKG, syntax="proto3")
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    _msg(fd, "Tokens", [("ids", 1, F.TYPE_UINT32, REP, None)])
    _msg(fd, "NextTokenChooserParameters", [
        ("temperature", 1, F.TYPE_FLOAT, OPT, None), ("top_k", 2, F.TYPE_UINT32, OPT, None), ("top_p", 3, F.TYPE_FLOAT, OPT, None),
        ("typical_p", 4, F.TYPE_FLOAT, OPT, None), ("do_sample", 5, F.TYPE_BOOL, OPT, None), ("seed", 6, F.TYPE_UINT64, OPT, None),
        ("repetition_penalty", 7, F.TYPE_FLOAT, OPT, None), ("presence_penalty", 8, F.TYPE_FLOAT, OPT, None),
        ("frequency_penalty", 9, F.TYPE_FLOAT, OPT, None), ("watermark", 10, F.TYPE_BOOL, OPT, None)])
    _msg(fd, "StoppingCriteriaParameters", [
        ("max_new_tokens", 1, F.TYPE_UINT32, OPT, None), ("stop_tokens", 2, F.TYPE_MESSAGE, OPT, "Tokens"),
        ("ignore_eos_token", 3, F.TYPE_BOOL, OPT, None)])
    _msg(fd, "Request", [
        ("id", 1, F.TYPE_UINT64, OPT, None), ("prompt", 2, F.TYPE_STRING, OPT, None), ("tokens", 3, F.TYPE_MESSAGE, OPT, "Tokens"),
        ("choosing_parameters", 4, F.TYPE_MESSAGE, OPT, "NextTokenChooserParameters"),
        ("stopping_parameters", 5, F.TYPE_MESSAGE, OPT, "StoppingCriteriaParameters")])
    _msg(fd, "BatchedRequest", [("req", 1, F.TYPE_MESSAGE, REP, "Request")])
    for ename, values in (("Status", ["PROCESSING", "FINISHED", "FAILED"]),
                          ("FinishReason", ["FINISH_REASON_LENGTH", "FINISH_REASON_EOS_TOKEN", "FINISH_REASON_STOP_SEQUENCE"])):
        e = fd.enum_type.add()
        e.name = ename
        for i, v in enumerate(values):
            ev = e.value.add()
            ev.name, ev.number = v, i
    _msg(fd, "Detail", [("logprobs", 1, F.TYPE_FLOAT, OPT, None), ("is_special", 2, F.TYPE_BOOL, OPT, None),
                        ("finish_reason", 3, F.TYPE_ENUM, OPT, "FinishReason")])
    _msg(fd, "Response", [("status", 1, F.TYPE_ENUM, OPT, "Status"), ("id", 2, F.TYPE_UINT64, OPT, None),
                          ("generated", 3, F.TYPE_STRING, OPT, None), ("tokens", 4, F.TYPE_MESSAGE, OPT, "Tokens"),
                          ("detail", 5, F.TYPE_MESSAGE, OPT, "Detail")])
    _msg(fd, "BatchedResponse", [("rsp", 1, F.TYPE_MESSAGE, REP, "Response")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return pool


_POOL = _build()


def _cls(name):
    return message_factory.GetMessageClass(_POOL.FindMessageTypeByName(f"{PKG}.{name}"))


Tokens = _cls("Tokens")
NextTokenChooserParameters = _cls("NextTokenChooserParameters")
StoppingCriteriaParameters = _cls("StoppingCriteriaParameters")
Request = _cls("Request")
BatchedRequest = _cls("BatchedRequest")
Detail = _cls("Detail")
Response = _cls("Response")
BatchedResponse = _cls("BatchedResponse")
PROCESSING, FINISHED, FAILED = 0, 1, 2
SERVICE = f"{PKG}.LLMService"
METHOD = f"/{SERVICE}/Generation"
