// csdr_compat.hpp -- the slice of the csdr 0.18 module API that digiham's operators are written against.
//
// The reference's public classes derive from Csdr::Module<T,U> / Csdr::AnyLengthModule<T,U>
// (include/rrc_filter.hpp:10, include/gfsk_demodulator.hpp:12, include/decoder.hpp:17) and talk to
// Csdr::Reader<T> / Csdr::Writer<U> (src/gfsk_demodulator/gfsk_demodulator.cpp:18-26,
// src/lib/decoder.cpp:21-32).  When the real csdr headers are installed they are used as they are;
// otherwise this header provides exactly the members the operators rely on, so the classes in this
// directory keep the reference's shape either way.
#pragma once

#if __has_include(<csdr/module.hpp>)
#include <csdr/module.hpp>
#include <csdr/reader.hpp>
#include <csdr/writer.hpp>
#include <csdr/ringbuffer.hpp>
#else

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace Csdr {

    template <typename T> class Reader {
        public:
            virtual ~Reader() = default;
            virtual size_t available() = 0;
            virtual T* getReadPointer() = 0;
            virtual void advance(size_t how_much) = 0;
    };

    template <typename T> class Writer {
        public:
            virtual ~Writer() = default;
            virtual size_t writeable() = 0;
            virtual T* getWritePointer() = 0;
            virtual void advance(size_t how_much) = 0;
    };

    template <typename T> class Sink {
        public:
            virtual ~Sink() = default;
            virtual void setReader(Reader<T>* r) { reader = r; }
            virtual Reader<T>* getReader() { return reader; }
        protected:
            Reader<T>* reader = nullptr;
    };

    template <typename T> class Source {
        public:
            virtual ~Source() = default;
            virtual void setWriter(Writer<T>* w) { writer = w; }
            virtual Writer<T>* getWriter() { return writer; }
        protected:
            Writer<T>* writer = nullptr;
    };

    template <typename T, typename U> class Module: public Sink<T>, public Source<U> {
        public:
            virtual bool canProcess() = 0;
            virtual void process() = 0;
        protected:
            std::mutex processMutex;
    };

    template <typename T, typename U> class AnyLengthModule: public Module<T, U> {
        public:
            bool canProcess() override {
                std::lock_guard<std::mutex> lock(this->processMutex);
                return std::min(this->reader->available(), this->writer->writeable()) > 0;
            }
            void process() override {
                std::lock_guard<std::mutex> lock(this->processMutex);
                size_t n = std::min(this->reader->available(), this->writer->writeable());
                process(this->reader->getReadPointer(), this->writer->getWritePointer(), n);
                this->reader->advance(n);
                this->writer->advance(n);
            }
        protected:
            virtual void process(T* input, U* output, size_t length) = 0;
    };

    // What the CLI driver needs (src/lib/cli.cpp:10,26-27,102-106): a buffer written by fread() and read by the
    // module through a RingbufferReader, and a writer that hands every advance() to stdout.  csdr's ring buffer
    // maps its memory twice to stay contiguous across the wrap; this one keeps a linear buffer and moves the unread
    // tail to the front when the write position reaches the end -- same contract (contiguous readable and writeable
    // regions), single reader.
    template <typename T> class RingbufferReader;

    template <typename T> class Ringbuffer: public Writer<T> {
        public:
            explicit Ringbuffer(size_t size): data(size) {}
            size_t writeable() override { compact(); return data.size() - tail; }
            T* getWritePointer() override { compact(); return data.data() + tail; }
            void advance(size_t how_much) override { tail += how_much; }
        private:
            friend class RingbufferReader<T>;
            void compact() {
                if (head == tail) { head = tail = 0; return; }
                if (tail == data.size() && head > 0) {
                    std::memmove(data.data(), data.data() + head, (tail - head) * sizeof(T));
                    tail -= head; head = 0;
                }
            }
            std::vector<T> data;
            size_t head = 0, tail = 0;
    };

    template <typename T> class RingbufferReader: public Reader<T> {
        public:
            explicit RingbufferReader(Ringbuffer<T>* buffer): buffer(buffer) {}
            size_t available() override { return buffer->tail - buffer->head; }
            T* getReadPointer() override { return buffer->data.data() + buffer->head; }
            void advance(size_t how_much) override { buffer->head += how_much; }
        private:
            Ringbuffer<T>* buffer;
    };

    template <typename T> class StdoutWriter: public Writer<T> {
        public:
            explicit StdoutWriter(size_t size = 262144): data(size) {}
            size_t writeable() override { return data.size(); }
            T* getWritePointer() override { return data.data(); }
            void advance(size_t how_much) override {
                fwrite(data.data(), sizeof(T), how_much, stdout);
                fflush(stdout);
            }
        private:
            std::vector<T> data;
    };

}

#endif
